#!/bin/bash
# A/B of non-temporal loads / stores in K6 (side builds with -DNEXTOU_K6_NT=n under tools/_ablate/, git-ignored).
# Build here (cross-compile): tools/k6_nt_ab.sh ; on the GPU box: tools/k6_nt_ab.sh run
set -e
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics"
if [ "$1" = "run" ]; then
  for n in 0 1 2 3; do
    echo "== NEXTOU_K6_NT=$n"; NEXTOU_HIP_LIB=$PWD/tools/_ablate/libnextou_hip_nt$n.so python tools/kernel_bench.py --norm --cl --iters 10 --only padded 2>&1 | grep -E "bn_cl|own fwd"
  done
  exit 0
fi
mkdir -p tools/_ablate
SRCS=$(python -c "from nextou_amd import build as b; import os; print(' '.join(os.path.join(b.CSRC, s) for s in b.HIP_SOURCES))")
for n in 0 1 2 3; do
  hipcc $FLAGS -DNEXTOU_K6_NT=$n -shared $SRCS -o tools/_ablate/libnextou_hip_nt$n.so &
done
# non-temporal loads in the two channels-last reductions only
hipcc $FLAGS -DNEXTOU_K6_NT_REDUCE=1 -shared $SRCS -o tools/_ablate/libnextou_hip_ntred.so &
wait
ls -la tools/_ablate | grep nt
