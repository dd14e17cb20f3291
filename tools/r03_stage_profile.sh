#!/bin/bash
# GNN blocks stage by stage (forward / forward+backward kernel time per module) + a short bench
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -x -q -k "norm or pad or tiny or channels_last" 2>&1 | tail -3
python tools/gnn_stage_profile.py > $OUT/gnn_stage_profile_all.txt 2>&1
grep -E "^s[0-9]|^sum" $OUT/gnn_stage_profile_all.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_padstage.json 2> $OUT/bench_padstage.log
python -c "import json;d=json.load(open('$OUT/bench_padstage.json'));print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'])"
