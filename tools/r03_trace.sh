#!/bin/bash
# per-kernel time inside the train step: rocprofv3 --kernel-trace of the eager bench step, fused pipeline on / off
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for mode in 1 0; do
  NEXTOU_PW_FUSE=$mode rocprofv3 --kernel-trace --stats -d $OUT/kt_fuse$mode -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 3 --no-cpu-baseline --graph off > $OUT/kt_fuse$mode.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $OUT/kt_fuse$mode $OUT/cfg2_step_kernel_trace_fuse$mode.md "cfg 2 step, eager, NEXTOU_PW_FUSE=$mode" --steady "knn_fused_kernel<28" 2
  find $OUT/kt_fuse$mode -name "*.db" -size +20M -delete
done
cd $GRAFT_REPO_ROOT
grep -E "pw_|bn_clw|igemm|Cijk" $OUT/cfg2_step_kernel_trace_fuse1.md | head -40
