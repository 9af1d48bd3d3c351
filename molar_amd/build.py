"""Builds libmolar_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so sits next to this
file so that it travels with the repo snapshot to the GPU box."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmolar_hip.so")
SOURCES = ["api.hip", "search.hip", "measure.hip", "pair_k0.hip", "pair_k1.hip", "pair_k2.hip", "pair_k3.hip"]
HEADERS = ["common.hpp", "boxmath.hpp", "linalg3.hpp", "pair_kernels.hpp", os.path.join("..", "..", "include", "molar_hip.h")]
# -ffp-contract=off: MolAR (Rust) never contracts a*b+c; bit-identical neighbour lists need the same
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in SOURCES:
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed: {' '.join(cmd)}\n{r.stdout}{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
