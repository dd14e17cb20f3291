"""Build recipe for libnextou_hip.so (gfx950 only) and the CPU oracle library.

``python -m nextou_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles for gfx950
without a GPU; the resulting ``.so`` lives in-tree (git-ignored) and travels with the snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libnextou_hip.so")

HIP_SOURCES = ["capi.hip", "knn_graph.hip", "mr_aggregate.hip", "bti_critical.hip", "norm_act.hip", "layout_ops.hip",
               "pw_gemm.hip", "head_rows.hip", "step_glue.hip", "stem_conv.hip"]
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off",                         # explicit fmaf only: bit-exact vs the oracle
    "-fhip-fp32-correctly-rounded-divide-sqrt",  # IEEE divide / sqrt in knn_prep
    "-munsafe-fp-atomics",                       # hardware f32 atomic add (coarse-grained HBM)
    "-Wall", "-Wno-unused-function",
]


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built on this machine")
    return exe


def build_hip(force: bool = False, verbose: bool = True) -> str:
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, "common.h"), os.path.join(REPO_DIR, "include", "nextou_hip.h"),
                   os.path.abspath(__file__)]
    if not force and _newer(LIB_PATH, deps):
        return LIB_PATH
    objs = []
    obj_dir = os.path.join(PKG_DIR, "csrc", "_obj")
    os.makedirs(obj_dir, exist_ok=True)
    procs = []
    for s in srcs:
        o = os.path.join(obj_dir, os.path.basename(s) + ".o")
        objs.append(o)
        if not force and _newer(o, [s] + deps[len(srcs):]):
            continue
        cmd = [hipcc()] + HIP_FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    print(build_hip(force="--force" in argv))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
