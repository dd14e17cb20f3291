#!/bin/bash
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
python -m pytest tests/test_gpu_fused.py -x -q --tb=short 2>&1 | tail -30 > $OUT/pytest_fused6.log; tail -8 $OUT/pytest_fused6.log
for v in "NEXTOU_PW_SO=0" "NEXTOU_PW_SO=1"; do
  echo "== $v"; env $v python tools/pw_gemm_probe.py --own-only --only "s2" --iters 10 2>&1 | grep -E "pw_wgrad" | tee -a $OUT/pw_wgrad_variants6.txt
done
for mode in 0 1; do
  NEXTOU_PW_FUSE=$mode python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_fuse6_$mode.json 2> $OUT/bench_fuse6_$mode.log
  python -c "import json;d=json.loads(open('$OUT/bench_fuse6_$mode.json').readline());print('NEXTOU_PW_FUSE=$mode', d['ms_per_step'], d['config']['step_replayed_as_hipgraph'], d['roofline']['own_kernels_ms_per_step'])"
done
NEXTOU_PW_FUSE=1 NEXTOU_PW_SO=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_fuse6_so0.json 2> $OUT/bench_fuse6_so0.log
python -c "import json;d=json.loads(open('$OUT/bench_fuse6_so0.json').readline());print('FUSE=1 SO=0', d['ms_per_step'])"
