#!/bin/bash
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
python -m pytest tests/test_gpu_fused.py -x -q --tb=short 2>&1 | tail -5 | tee $OUT/pytest_fused7.log
for lg in default 3 4 5 6; do
  echo "== NEXTOU_CLW_LG=$lg"
  if [ $lg = default ]; then python tools/kernel_bench.py --norm --cl --iters 10 --only "s2" 2>&1 | grep -E "bn_clw" ; else NEXTOU_CLW_LG=$lg python tools/kernel_bench.py --norm --cl --iters 10 --only "s2" 2>&1 | grep -E "bn_clw"; fi
done | tee $OUT/clw_lg_sweep.txt
python tools/pw_gemm_probe.py --own-only --only "FFN s2 132" --iters 10 2>&1 | grep -E "pw_" | head
python tools/gnn_stage_profile.py --cl --iters 10 --only "s2 Pool FFN" --kernels 2>&1 | grep -E "^s2|pw_|bn_" | cut -c1-140
