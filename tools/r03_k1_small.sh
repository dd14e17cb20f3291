#!/bin/bash
# K1 prep / merge rework: parity tests, then per-kernel times at the cfg-2 call shapes, old vs new
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py tests/test_gpu_property.py -x -q -k "knn or pairwise or edge" 2>&1 | tail -3
python tools/kernel_bench.py --cfg 2 --iters 10 2>&1 | grep -E "knn_|total own" > $OUT/k1_small_new.txt
NEXTOU_KNN_PREP=v1 NEXTOU_KNN_MERGE=v1 python tools/kernel_bench.py --cfg 2 --iters 10 2>&1 | grep -E "knn_|total own" > $OUT/k1_small_old.txt
python tools/kernel_bench.py --cfg 5 --iters 5 2>&1 | grep -E "knn_prep|knn_merge|total own" > $OUT/k1_small_new_cfg5.txt
NEXTOU_KNN_PREP=v1 NEXTOU_KNN_MERGE=v1 python tools/kernel_bench.py --cfg 5 --iters 5 2>&1 | grep -E "knn_prep|knn_merge|total own" > $OUT/k1_small_old_cfg5.txt
paste -d'\n' $OUT/k1_small_old.txt /dev/null | cut -c1-120
echo ======== new; cut -c1-120 $OUT/k1_small_new.txt
echo ======== cfg5 old; cut -c1-120 $OUT/k1_small_old_cfg5.txt; echo ======== cfg5 new; cut -c1-120 $OUT/k1_small_new_cfg5.txt
