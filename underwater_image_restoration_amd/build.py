"""Build libuir_hip.so (all HIP kernels + the C ABI of include/uir_kernels.h) for gfx950.

In-tree build: objects under csrc/_build/, the shared library next to this file, so the
library travels with the source tree to the GPU box.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "libuir_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
SOURCES = ["gemm", "tiled", "gram", "dwconv", "gdfn", "layernorm", "attn", "spectral", "shuffle", "resize", "colsum", "loss", "ssim", "uiqm", "frfn", "tokenfft", "winattn", "optim", "abi_layout"]


def source_sha() -> str:
    """sha256 (16 hex) over everything that decides which kernels a train step runs and how: csrc/*.hip, *.h, the C header and
    the launch-plan compiler.  Measurement files (profiles/*pmc_traffic*.json) carry it; bench.py refuses a stale one."""
    import hashlib
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files += [os.path.join(HERE, "..", "include", "uir_kernels.h"), os.path.join(HERE, "engine.py")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _deps():
    return [os.path.join(CSRC, "uir_common.h"), os.path.join(HERE, "..", "include", "uir_kernels.h")]


def _stale(target, srcs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def _compile(name):
    src = os.path.join(CSRC, name + ".hip")
    obj = os.path.join(OBJ, name + ".o")
    if _stale(obj, [src] + _deps()):
        subprocess.run([HIPCC, *FLAGS, "-c", src, "-o", obj], check=True)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for n in SOURCES:
            p = os.path.join(OBJ, n + ".o")
            if os.path.exists(p):
                os.remove(p)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if force or _stale(LIB, objs):
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], check=True)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
