#!/bin/bash
# fused point-wise pipeline: parity tests, then the step A/B (NEXTOU_PW_FUSE = 0 | fwd | 1) in one call on one box
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -15 | tee $OUT/pytest_fused.log
python -m pytest tests/test_gpu_parity2.py tests/test_gpu_parity.py -x -q -k "pointwise or blocks or ffn or tiny or graphed or channel_padding or layout_policies" 2>&1 | tail -8 | tee $OUT/pytest_fused_related.log
for mode in 0 fwd 1; do
  NEXTOU_PW_FUSE=$mode python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_fuse_$mode.json 2> $OUT/bench_fuse_$mode.log
  python -c "import json;d=json.loads(open('$OUT/bench_fuse_$mode.json').readline());print('NEXTOU_PW_FUSE=$mode', d['ms_per_step'], d['config']['step_replayed_as_hipgraph'], d['roofline']['own_kernels_ms_per_step'])"
done
NEXTOU_PW_FUSE=0 NEXTOU_PW_GEMM=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_fuse_0_k7.json 2> $OUT/bench_fuse_0_k7.log
python -c "import json;d=json.loads(open('$OUT/bench_fuse_0_k7.json').readline());print('NEXTOU_PW_FUSE=0 PW_GEMM=1', d['ms_per_step'])"
python tools/gnn_stage_profile.py --cl --iters 10 > $OUT/gnn_stage_profile_fused.txt 2>&1; tail -20 $OUT/gnn_stage_profile_fused.txt
NEXTOU_PW_FUSE=0 python tools/gnn_stage_profile.py --cl --iters 10 > $OUT/gnn_stage_profile_unfused.txt 2>&1; tail -3 $OUT/gnn_stage_profile_unfused.txt
