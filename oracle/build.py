"""Builds oracle/liboracle.so from oracle/nextou_oracle.c (gcc).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "nextou_oracle.c")
LIB = os.path.join(HERE, "liboracle.so")
# -ffp-contract=off: every rounding stays where the source writes it; -mfma only makes fmaf() one
# instruction (x86-64-v3 is safe on any host this runs on); no -ffast-math, no -march=native.
FLAGS = ["-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-mavx2", "-mfma",
         "-fopenmp", "-Wall"]


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(
            os.path.getmtime(SRC), os.path.getmtime(os.path.abspath(__file__))):
        return LIB
    cmd = ["gcc"] + FLAGS + [SRC, "-o", LIB, "-lm"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
