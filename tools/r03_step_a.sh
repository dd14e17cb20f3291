#!/bin/bash
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
python -m pytest tests/test_gpu_fused.py -x -q --tb=short 2>&1 | tail -30 > $OUT/pytest_fused5.log; tail -8 $OUT/pytest_fused5.log
for v in "NEXTOU_PW_SW=0" "NEXTOU_PW_SW=2"; do
  echo "== $v"; env $v python tools/pw_gemm_probe.py --own-only --only "s2" --iters 10 2>&1 | grep -E "pw_rows" | tee -a $OUT/pw_rows_variants5.txt
done
for mode in 0 1; do
  NEXTOU_PW_FUSE=$mode python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_fuse5_$mode.json 2> $OUT/bench_fuse5_$mode.log
  python -c "import json;d=json.loads(open('$OUT/bench_fuse5_$mode.json').readline());print('NEXTOU_PW_FUSE=$mode', d['ms_per_step'], d['config']['step_replayed_as_hipgraph'], d['roofline']['own_kernels_ms_per_step']); print(d['roofline_graph']['K7_pointwise_rows'])"
done
NEXTOU_PW_FUSE=1 NEXTOU_PW_SW=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_fuse5_sw0.json 2> $OUT/bench_fuse5_sw0.log
python -c "import json;d=json.loads(open('$OUT/bench_fuse5_sw0.json').readline());print('FUSE=1 SW=0', d['ms_per_step'])"
python tools/gnn_stage_profile.py --cl --iters 10 --only s2 --kernels > $OUT/gnn_stage_profile_fused5.txt 2>&1; grep -E "^s2|pw_|bn_|sum over" $OUT/gnn_stage_profile_fused5.txt | cut -c1-150
