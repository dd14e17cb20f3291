#!/bin/bash
# checkpoint: full GPU test suite (cfg-5 test apart, with its phase timings), smoke, bench
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
python -m pytest tests -m gpu -q -x --durations=12 --deselect tests/test_gpu_cfg5.py 2>&1 | grep -v GridwiseOp | tail -25 > $OUT/pytest_gpu_checkpoint2.log; tail -22 $OUT/pytest_gpu_checkpoint2.log
python -m pytest tests/test_gpu_cfg5.py -q -x -s 2>&1 | grep -E "cfg5|oracle checks|passed|failed" | tail -8 > $OUT/pytest_cfg5_phases.log; cat $OUT/pytest_cfg5_phases.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --no-cpu-baseline > $OUT/bench_checkpoint2.json 2> $OUT/bench_checkpoint2.log; cat $OUT/bench_checkpoint2.json | cut -c1-600
