#!/bin/bash
# Which part of pw_rows_kernel bounds it?  Side builds with -DNEXTOU_PW_ABLATE=n under tools/_ablate/ (git-ignored):
# 1 = no MFMAs, 2 = no global loads / LDS stores after a tile's first stage, 4 = no result stores, and combinations.
# Build here: tools/pw_ablate.sh ; on the GPU box: tools/pw_ablate.sh run
set -e
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics"
VARIANTS="${PW_VARIANTS-1 2 4 6 3}"
if [ "$1" = "run" ]; then
  echo "== product build"; python tools/pw_gemm_probe.py --own-only --only "FFN s2" --iters 5 2>&1 | grep "pw_rows" | cut -c1-110
  for n in $VARIANTS; do
    echo "== NEXTOU_PW_ABLATE=$n"; NEXTOU_HIP_LIB=$PWD/tools/_ablate/libnextou_hip_pw$n.so python tools/pw_gemm_probe.py --own-only --only "FFN s2" --iters 5 2>&1 | grep "pw_rows" | cut -c1-110
  done
  for n in 2 4 6 9; do
    echo "== NEXTOU_PW_STAGGER=$n"; NEXTOU_HIP_LIB=$PWD/tools/_ablate/libnextou_hip_pwst$n.so python tools/pw_gemm_probe.py --own-only --only "FFN s2" --iters 5 2>&1 | grep "pw_rows" | cut -c1-110
  done
  exit 0
fi
mkdir -p tools/_ablate
SRCS=$(python -c "from nextou_amd import build as b; import os; print(' '.join(os.path.join(b.CSRC, s) for s in b.HIP_SOURCES))")
for n in $VARIANTS; do
  hipcc $FLAGS -DNEXTOU_PW_ABLATE=$n -shared $SRCS -o tools/_ablate/libnextou_hip_pw$n.so &
done
for n in 2 4 6 9; do
  hipcc $FLAGS -DNEXTOU_PW_STAGGER=$n -shared $SRCS -o tools/_ablate/libnextou_hip_pwst$n.so &
done
wait
ls -la tools/_ablate | grep pw
