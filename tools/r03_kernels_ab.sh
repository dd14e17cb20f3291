#!/bin/bash
# round 3, second kernel batch: parity of the new K2 (K-split quad) / K1 (window plan) / K5 (separable critical map, register CE)
# kernels and of the fused pipeline, then their kernel benches and the K7 rows experiments
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -6 | tee $OUT/pytest_fused2.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -x -q -k "knn or mr_ or bti or critical or ti_loss or compound or gather or public" 2>&1 | tail -6 | tee $OUT/pytest_kernels2.log
python tools/kernel_bench.py --cfg 2 --iters 10 > $OUT/kernel_bench_cfg2_v2.txt 2>&1; grep -E "mr_fwd|knn_fused|knn_merge|total" $OUT/kernel_bench_cfg2_v2.txt
NEXTOU_MR_FWD=v1 NEXTOU_KNN_WINDOW=0 python tools/kernel_bench.py --cfg 2 --iters 10 > $OUT/kernel_bench_cfg2_v2_old.txt 2>&1; grep -E "mr_fwd|knn_fused|knn_merge|total" $OUT/kernel_bench_cfg2_v2_old.txt
python tools/kernel_bench.py --cfg 5 --iters 5 > $OUT/kernel_bench_cfg5_v2.txt 2>&1; grep -E "mr_fwd|total" $OUT/kernel_bench_cfg5_v2.txt
python tools/kernel_bench.py --bti --iters 10 --json $OUT/kernel_bench_k5_v2.json > $OUT/kernel_bench_k5_v2.txt 2>&1; tail -22 $OUT/kernel_bench_k5_v2.txt
for v in "NEXTOU_PW_KTAIL=0" "NEXTOU_PW_KTAIL=1" "NEXTOU_PW_KTAIL=1 NEXTOU_PW_ROWS_TM=1"; do
  echo "== $v"; env $v python tools/pw_gemm_probe.py --own-only --only "s2" --iters 10 2>&1 | grep -E "pw_rows_kernel" | tee -a $OUT/pw_rows_variants.txt
done
