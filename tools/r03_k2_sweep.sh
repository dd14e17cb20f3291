#!/bin/bash
# K2 forward (mr_fwd_qb_kernel) plan sweep at the cfg-2 call shapes: quads x threads x grid size, then SQ counters.
#   on the GPU box:  tools/r03_k2_sweep.sh [tests]    -> gpurun_out/r03/k2_sweep_qb.txt, k2_qb_sq.md
# (the ablated side builds of the K-split kernel this replaced are recorded in profiles/r03_k2_pooled_forward.md)
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
if [ "$1" = "tests" ]; then
  python -m pytest tests/test_gpu_parity.py -q -x -k "mr_forward_kernel_variants or mr_aggregate" 2>&1 | tail -5
fi
R=$OUT/k2_sweep_qb.txt
: > $R
run() { python tools/kernel_bench.py --cfg 2 --iters 10 --only "$1" 2>&1 | grep -E "mr_fwd" ; }
for shape in "s3 Pool" "s4 Pool" "s5 Pool" "s2 Pool" "s3 Swin" "s2 Swin"; do
  echo "== $shape: default dispatch" >> $R
  run "$shape" >> $R
  echo "== $shape: mr_fwd_qb_kernel plan sweep (quads, threads, target WGs)" >> $R
  for q in 1 2 4; do for t in 256 512; do for w in 700 1024 1500 2300 3000 4600; do
    echo "-- quads $q threads $t wgs $w" >> $R
    NEXTOU_QB_QUADS=$q NEXTOU_QB_THREADS=$t NEXTOU_QB_WGS=$w run "$shape" >> $R
  done; done; done
done
python - <<'PY'
import re
cur = None
for l in open("gpurun_out/r03/k2_sweep_qb.txt"):
    l = l.rstrip()
    if l.startswith("=="): print(l); continue
    if l.startswith("--"): cur = l; continue
    if "mr_fwd" in l:
        f = l.split(); i = f.index("hbm")
        print("%-34s %-14s %s us" % (cur or "", f[2][:14], f[i + 1]))
PY
bash tools/pmc_sq.sh "s3 Pool" "" k2_qb_pool_s3 > /dev/null 2>&1
python tools/pmc_table.py gpurun_out/sq_k2_qb_pool_s3/g1 gpurun_out/sq_k2_qb_pool_s3/g2 gpurun_out/sq_k2_qb_pool_s3/g3 --match mr_fwd > $OUT/k2_qb_sq.md 2>&1; cat $OUT/k2_qb_sq.md
