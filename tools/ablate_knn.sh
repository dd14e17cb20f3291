#!/bin/bash
# Builds ablated side copies of libnextou_hip.so (tools/_ablate/, git-ignored) for the kNN kernel:
#   1 = no top-K pushes, 2 = no slab staging after the first, 4 = no MFMA, and combinations.
# Run here (cross-compile), then on the GPU box:  tools/ablate_knn.sh run
set -e
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics"
if [ "$1" = "run" ]; then
  for n in ${ABLATE_SET:-0 1 2 4 3 5 6}; do
    echo "== ablate $n"; NEXTOU_HIP_LIB=$PWD/tools/_ablate/libnextou_hip_a$n.so python tools/kernel_bench.py --cfg 2 2>&1 | grep -E "knn_fused|knn_merge"
  done
  exit 0
fi
mkdir -p tools/_ablate
# the source list comes from nextou_amd/build.py, so a side library exports every symbol _lib.py binds
SRCS=$(python -c "from nextou_amd import build as b; import os; print(' '.join(os.path.join(b.CSRC, s) for s in b.HIP_SOURCES))")
for n in ${ABLATE_SET:-0 1 2 4 3 5 6}; do
  hipcc $FLAGS -DNEXTOU_ABLATE=$n -shared $SRCS -o tools/_ablate/libnextou_hip_a$n.so &
done
wait
ls -la tools/_ablate
