#!/bin/bash
OUT=$PWD/gpurun_out/r04
mkdir -p $OUT
export MIOPEN_LOG_LEVEL=1
python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_parity2.py -x -q -k "norm or pad" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg2_k6pad.json 2> $OUT/bench_cfg2_k6pad.log
NEXTOU_K6_SKIP_PAD=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg2_k6nopad.json 2> $OUT/bench_cfg2_k6nopad.log
for f in bench_cfg2_k6pad bench_cfg2_k6nopad; do python -c "import json;d=json.load(open('$OUT/$f.json'));r=d['roofline'];print('$f', d['ms_per_step'], r['kernel'], r['avg_us'], r['frac'], r['frac_on_unpadded_bytes'], r['own_kernels_ms_per_step'])"; done
python tools/gnn_stage_profile.py --cl --kernels --stages 2 --iters 5 > $OUT/gnn_stage_s2_kernels.txt 2>&1
NEXTOU_KNN_TILES=4 python tools/kernel_bench.py --cfg 2 --only "s3 Pool" > $OUT/kernel_bench_s3pool_tiles4.txt 2>&1
python tools/kernel_bench.py --cfg 2 --only "s3 Pool" > $OUT/kernel_bench_s3pool_tiles2.txt 2>&1
grep -h knn_ $OUT/kernel_bench_s3pool_tiles4.txt $OUT/kernel_bench_s3pool_tiles2.txt
