#!/bin/bash
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -q -x -k "layout_policy" 2>&1 | tail -2
python tools/gnn_stage_profile.py --cl --iters 10 --kernels > $OUT/gnn_stage_profile_cl_kernels.txt 2>&1
grep -E "^s[0-9]|^sum" $OUT/gnn_stage_profile_cl_kernels.txt
