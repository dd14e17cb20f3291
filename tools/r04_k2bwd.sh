#!/bin/bash
# K2 backward: 64-bit fixed-point LDS accumulators (default) vs the float-atomic scatter (NEXTOU_MR_BWD=float)
OUT=$PWD/gpurun_out/r04
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py tests/test_gpu_cfg5.py -x -q -k "mr_ or mrconv or gather or cfg5 or blocks" 2>&1 | tail -5
python tools/kernel_bench.py --cfg 2 > $OUT/kernel_bench_cfg2_fix.txt 2>&1
NEXTOU_MR_BWD=float python tools/kernel_bench.py --cfg 2 > $OUT/kernel_bench_cfg2_floatatomics.txt 2>&1
python tools/kernel_bench.py --cfg 5 > $OUT/kernel_bench_cfg5_fix.txt 2>&1
NEXTOU_MR_BWD=float python tools/kernel_bench.py --cfg 5 > $OUT/kernel_bench_cfg5_floatatomics.txt 2>&1
grep -h "mr_bwd" $OUT/kernel_bench_cfg2_fix.txt $OUT/kernel_bench_cfg2_floatatomics.txt $OUT/kernel_bench_cfg5_fix.txt $OUT/kernel_bench_cfg5_floatatomics.txt
python -m pytest tests/test_gpu_fused_goldens.py -x -q 2>&1 | tail -5
