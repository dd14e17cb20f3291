#!/bin/bash
# grouped 1x1 rows kernel: parity, then stage-2 Swin grapher per-kernel profile and a short bench, new vs NEXTOU_PW_GRP=0
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_parity2.py -x -q -k "pointwise or pw_ or k7" 2>&1 | tail -2
python tools/pw_gemm_probe.py 2>&1 | grep -E "gconv|g6" | cut -c1-160 > $OUT/pw_grp_probe.txt; cat $OUT/pw_grp_probe.txt
NEXTOU_PW_GRP=0 python tools/pw_gemm_probe.py 2>&1 | grep -E "gconv|g6" | cut -c1-160 > $OUT/pw_grp_probe_off.txt; cat $OUT/pw_grp_probe_off.txt
python tools/gnn_stage_profile.py --cl --iters 10 --only "s2 Swin" --kernels 2>&1 | grep -E "^s2|pw_rows" | cut -c1-150 > $OUT/gnn_s2_swin_grp.txt; cat $OUT/gnn_s2_swin_grp.txt
for mode in 1 0; do
  NEXTOU_PW_GRP=$mode python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_grp$mode.json 2> $OUT/bench_grp$mode.log
  python -c "import json;d=json.load(open('$OUT/bench_grp$mode.json'));print('NEXTOU_PW_GRP=$mode', d['ms_per_step'], d['value'])"
done
