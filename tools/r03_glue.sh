#!/bin/bash
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
python -m pytest tests/test_gpu_inference.py tests/test_gpu_cfg5.py -x -q --tb=short --durations=5 2>&1 | grep -v "GridwiseOp" | tail -14 > $OUT/pytest_newtests2.log; tail -12 $OUT/pytest_newtests2.log
python tools/aten_glue_profile.py --top 40 --ops copy_,cat,add,fill_,zero_,mul,clone,contiguous,sum,pad,constant_pad > $OUT/aten_glue_shapes.txt 2>&1; grep -E "^\|" $OUT/aten_glue_shapes.txt | head -48 | cut -c1-250
python tools/aten_glue_profile.py --top 30 --stacks --ops copy_,cat,add,fill_,zero_,mul,pad > $OUT/aten_glue_stacks.txt 2>&1; grep -E "^\|" $OUT/aten_glue_stacks.txt | head -34 | cut -c1-330
