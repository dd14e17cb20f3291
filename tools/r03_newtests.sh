#!/bin/bash
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
python -m pytest tests/test_gpu_inference.py tests/test_gpu_cfg5.py -x -q --tb=short 2>&1 | grep -v "GridwiseOp" | tail -30 > $OUT/pytest_newtests.log; tail -25 $OUT/pytest_newtests.log
python -m pytest tests/test_gpu_parity.py -x -q --tb=short -k "two_ranks" 2>&1 | grep -v "GridwiseOp" | tail -5
