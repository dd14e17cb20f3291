#!/bin/bash
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
python -m pytest tests -m gpu -x -q --tb=short 2>&1 | grep -v "GridwiseOp\|^$" | tail -15 > $OUT/pytest_gpu_checkpoint1.log; tail -6 $OUT/pytest_gpu_checkpoint1.log
for mode in 1 0; do
  NEXTOU_PW_FUSE=$mode python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_ck1_$mode.json 2> $OUT/bench_ck1_$mode.log
  python -c "import json;d=json.loads(open('$OUT/bench_ck1_$mode.json').readline());print('NEXTOU_PW_FUSE=$mode', d['ms_per_step'], d['config']['step_replayed_as_hipgraph'], d['roofline']['own_kernels_ms_per_step'], d['roofline']['frac'])"
done
python tools/kernel_bench.py --norm --cl --iters 10 > $OUT/kernel_bench_norm_cl_v2.txt 2>&1; grep -E "bn_clw|own fwd" $OUT/kernel_bench_norm_cl_v2.txt | cut -c1-150
