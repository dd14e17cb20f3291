#!/bin/bash
# HBM traffic of the dominant kernels from the PMC counters, as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC has 4 slots; FETCH_SIZE takes 3, WRITE_SIZE 2),
# counters only with --kernel-trace.  Run on the GPU box from the repo root:  tools/pmc_traffic.sh "s3 Pool"
# (second argument: extra tools/kernel_bench.py flags, e.g. --norm for the K6 calls)
set -e
ONLY="${1:-s3 Pool}"
EXTRA="${2:-}"
TAG=$(echo "$ONLY" | tr " " "_")
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o pmc -- python $GRAFT_REPO_ROOT/tools/kernel_bench.py --cfg 2 --iters 3 $EXTRA --only "$ONLY" > $OUT/$c.log 2>&1 || tail -5 $OUT/$c.log
done
find $OUT -name "*.csv" | head
