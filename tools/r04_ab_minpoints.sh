#!/bin/bash
# fused chain at every stage (MIN_POINTS=0) vs the 65 536-point threshold, eager and as replayed hipGraphs
OUT=$PWD/gpurun_out/r04
mkdir -p $OUT
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
export MIOPEN_LOG_LEVEL=1
NEXTOU_PW_FUSE_MIN_POINTS=0 python tools/gnn_stage_profile.py --cl --graph --iters 20 --stages 3,4,5 > $OUT/gnn_stage_min0.txt 2>&1
grep -E "^s[0-9]|^sum|^as " $OUT/gnn_stage_min0.txt
NEXTOU_PW_FUSE_MIN_POINTS=0 python tools/gnn_stage_profile.py --cl --kernels --stages 3,4,5 --iters 5 > $OUT/gnn_stage_min0_kernels.txt 2>&1
python -m pytest tests/test_gpu_parity.py -x -q -k "argmax or bti or label" 2>&1 | tail -3
