#!/bin/bash
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
python -m pytest tests/test_gpu_fused.py -q -x -k "cross_entropy" 2>&1 | tail -2
python tools/loss_probe.py 2>&1 | grep -E "loss, forward|grad layouts|ce_mean" | cut -c1-200
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed" | tail -3
for v in 1 0; do
  NEXTOU_FUSED_CE=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_ce2_$v.json 2> $OUT/bench_ce2_$v.log
  python -c "import json;d=json.load(open('$OUT/bench_ce2_$v.json'));print('NEXTOU_FUSED_CE=$v', d['ms_per_step'], d['config']['final_loss'])"
done
