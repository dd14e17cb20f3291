#!/bin/bash
# round 3 kernel batch: parity of the new K2 / K1 / K5 / K7 kernels and of the fused pipeline, their kernel benches, K7 variants
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
python -m pytest tests/test_gpu_fused.py -x -q --tb=short 2>&1 | tail -40 > $OUT/pytest_fused3.log; tail -25 $OUT/pytest_fused3.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -x -q --tb=short -k "knn or mr_ or bti or critical or ti_loss or compound or gather or public or pointwise" 2>&1 | tail -12 | tee $OUT/pytest_kernels3.log
python tools/kernel_bench.py --cfg 2 --iters 10 > $OUT/kernel_bench_cfg2_v3.txt 2>&1; grep -E "mr_fwd|knn_fused|knn_merge|total" $OUT/kernel_bench_cfg2_v3.txt
python tools/kernel_bench.py --cfg 5 --iters 5 > $OUT/kernel_bench_cfg5_v3.txt 2>&1; grep -E "mr_fwd|total" $OUT/kernel_bench_cfg5_v3.txt
python tools/kernel_bench.py --bti --iters 10 --json $OUT/kernel_bench_k5_v3.json > $OUT/kernel_bench_k5_v3.txt 2>&1; grep -E "critical_kernel|total" $OUT/kernel_bench_k5_v3.txt
for v in "NEXTOU_PW_SW=0" "NEXTOU_PW_SW=1"; do
  echo "== $v"; env $v python tools/pw_gemm_probe.py --own-only --only "s2" --iters 10 2>&1 | grep -E "pw_rows" | tee -a $OUT/pw_rows_variants3.txt
done
for mode in 0 1; do
  NEXTOU_PW_FUSE=$mode python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_fuse3_$mode.json 2> $OUT/bench_fuse3_$mode.log
  python -c "import json;d=json.loads(open('$OUT/bench_fuse3_$mode.json').readline());print('NEXTOU_PW_FUSE=$mode', d['ms_per_step'], d['config']['step_replayed_as_hipgraph'], d['roofline']['own_kernels_ms_per_step'])"
done
