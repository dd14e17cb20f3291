#!/bin/bash
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
python -m pytest tests/test_gpu_fused.py -q -x -k "rows_gemm" 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py tests/test_gpu_inference.py -q -x -k "forward_parity or tiny or train_step or blocks or inference or attribution or ffn" 2>&1 | tail -3
python tools/gnn_stage_profile.py --cl --iters 10 > $OUT/gnn_stage_profile_mm.txt 2>&1; grep -E "^s[345]|^sum" $OUT/gnn_stage_profile_mm.txt
NEXTOU_PW_MM_MAX_POINTS=0 python tools/gnn_stage_profile.py --cl --iters 10 > $OUT/gnn_stage_profile_mm0.txt 2>&1; grep -E "^s[345]|^sum" $OUT/gnn_stage_profile_mm0.txt
for v in 8192 0 8192 0; do
  NEXTOU_PW_MM_MAX_POINTS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_mm_$v.json 2> $OUT/bench_mm_$v.log
  python -c "import json;d=json.load(open('$OUT/bench_mm_$v.json'));print('NEXTOU_PW_MM_MAX_POINTS=$v', d['ms_per_step'])"
done
