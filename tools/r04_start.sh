#!/bin/bash
# round-4 opening measurement: bench line + the GNN blocks eager / hipGraph replay / per kernel
OUT=$PWD/gpurun_out/r04
mkdir -p $OUT
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
export MIOPEN_LOG_LEVEL=1
python tools/gnn_stage_profile.py --cl --graph --iters 20 > $OUT/gnn_stage_start.txt 2>&1
grep -E "^s[0-9]|^sum|^as " $OUT/gnn_stage_start.txt
python tools/gnn_stage_profile.py --cl --kernels --stages 3,4,5 --iters 5 > $OUT/gnn_stage_start_kernels.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_start.json 2> $OUT/bench_start.log
python -c "import json;d=json.load(open('$OUT/bench_start.json'));print(d['ms_per_step'], d['value'])"
