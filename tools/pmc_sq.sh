#!/bin/bash
# SQ-level counters for the own kernels (one rocprofv3 --pmc pass per counter group, --kernel-trace only).
# usage on the GPU box from the repo root: tools/pmc_sq.sh "s2 Swin" ["--norm --cl" [tag]]
# (second argument: extra tools/kernel_bench.py flags; third: directory tag, default derived from the first)
ONLY="${1:-s2 Swin}"
EXTRA="${2:-}"
TAG="${3:-$(echo "$ONLY" | tr " " "_")}"
OUT=$PWD/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o pmc -- python $GRAFT_REPO_ROOT/tools/kernel_bench.py --cfg 2 --iters 3 $EXTRA --only "$ONLY" > $OUT/g$i.log 2>&1 || tail -3 $OUT/g$i.log
done
ls $OUT
