#!/bin/bash
# K1 on the <= 192-point graphs (cfg-2 stages 3-5) and the stage-4 pooled graph: sweep of the fused kernel's decomposition
# (waves per workgroup, chunk width, channels per slab, candidate splits) through the NEXTOU_KNN_* experiment switches.
OUT=$PWD/gpurun_out/${1:-r04}
mkdir -p $OUT
F=$OUT/k1_small_sweep.txt
: > $F
run() {
  echo "## $*" >> $F
  env "$@" python tools/kernel_bench.py --cfg 2 --iters 10 2>/dev/null | grep -E "^s[345] .*knn_" >> $F
}
run X=default
for nw in 1 2 3; do for tiles in 1 2; do for ks in 64 128; do
  run NEXTOU_KNN_NW=$nw NEXTOU_KNN_TILES=$tiles NEXTOU_KNN_KS=$ks
done; done; done
run NEXTOU_KNN_TILES=4
run NEXTOU_KNN_NW=2 NEXTOU_KNN_TILES=6
run NEXTOU_KNN_NW=1 NEXTOU_KNN_TILES=2 NEXTOU_KNN_KS=64 NEXTOU_KNN_SPLITS=1
cat $F
