"""Builds the gfx950 shared libraries in-tree (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libpais_hip.so")

HIP_SOURCES = ["pais_kernels.hip", "pais_capi.hip", "pais_mvs.hip", "pais_io.hip", "pais_pyramid.hip", "pais_seed.hip"]  # pais_mvs.hip: host scheduler
HEADERS = ["pais_dev.hpp", "pais_detmath.hpp", "pais_internal.h", "pais_eval.hpp", "pais_tile.hpp", "pais_tile2.hpp", "pais_literal.hpp", "pais_pre.hpp", os.path.join("..", "..", "include", "pais_hip.h"),
           os.path.join("..", "..", "include", "pais_mvs.h"), os.path.join("..", "..", "include", "pais_io.h"),
           os.path.join("..", "..", "include", "pais_pyramid.h"), os.path.join("..", "..", "include", "pais_seed.h")]
# -ffp-contract=off: the PSO position/velocity update and the per-tap arithmetic keep
# the reference's rounding sequence; measured cost is reported there.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-Wno-unused-result"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the PAIS HIP library cannot be built (there is no CPU fallback)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in HIP_SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, f) for f in HIP_SOURCES if os.path.exists(os.path.join(CSRC, f))]
    cmd = [hipcc()] + HIPCC_FLAGS + srcs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
