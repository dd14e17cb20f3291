#!/bin/bash
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
for v in 1 0 1 0; do
  NEXTOU_CAT_BIAS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cat_$v.json 2> $OUT/bench_cat_$v.log
  python -c "import json;d=json.load(open('$OUT/bench_cat_$v.json'));print('NEXTOU_CAT_BIAS=$v', d['ms_per_step'], d['config']['final_loss'])"
done
