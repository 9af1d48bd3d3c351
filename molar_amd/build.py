"""Builds libmolar_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so sits next to this
file so that it travels with the repo snapshot to the GPU box."""
from __future__ import annotations

import fcntl
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmolar_hip.so")
SOURCES = ["api.hip", "search.hip", "search_f64.hip", "measure.hip", "measure_f64.hip", "membrane.hip", "xtc.hip", "pair_k0.hip", "pair_k1.hip", "pair_k2.hip", "pair_k3.hip", "pair_k4.hip", "pair_k5.hip", "pair_k6.hip", "pair_k7.hip", "pair_k8.hip", "pair_small.hip", "devsort.hip"]
HEADERS = ["common.hpp", "boxmath.hpp", "linalg3.hpp", "pair_kernels.hpp", "hist_kernels.hpp", "hoststream.hpp", "boxmath64.hpp", "stages.hpp", os.path.join("..", "..", "include", "molar_hip.h")]
# -ffp-contract=off: MolAR (Rust) never contracts a*b+c; bit-identical neighbour lists need the same
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function",
         # SLP vectorisation packs the x/z differences of ONE candidate into v_pk_add/v_pk_mul_f32, which issue at
         # half rate on gfx950 and need extra v_mov to line up their operands
         "-fno-slp-vectorize"]
# experiments: extra compiler flags, e.g. MOLAR_HIP_EXTRA_FLAGS="-mllvm -amdgpu-sched-strategy=max-ilp"
FLAGS += os.environ.get("MOLAR_HIP_EXTRA_FLAGS", "").split()


HASH_FILE = LIB + ".srchash"


def _source_hash() -> str:
    """Content hash of everything the library is built from (file mtimes do not survive a repo snapshot)."""
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for name in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(HASH_FILE):
        return True
    try:
        return open(HASH_FILE).read().strip() != _source_hash()
    except OSError:
        return True


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    # several ranks of one node may get here at once (torch.distributed.run): build under a lock
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB
            return _build_locked(verbose, force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def exec_restore_hazards(asm_path: str) -> list[str]:
    """Audit of generated gfx950 ISA for one register-allocator hazard of ROCm 7.2's compiler: a live-range split of a
    VGPR whose copy (`v_mov vA, vB`, later undone by `v_mov vB, vA`) or spill store was placed at the top of a join block
    AHEAD of the instruction that restores EXEC (`s_or_b64 exec, exec, ...`).  Such a copy moves only the lanes that were
    active inside the divergent region; when the value is live for all lanes (a wave-uniform value kept in a VGPR, a lane
    id) the masked-off lanes keep a stale copy.  It happened in hist_kernel (round 2, found by tools/fuzz_search.py: wrong
    histogram bins for the lanes concerned); the build refuses a library whose kernels contain the pattern.  (A copy in
    front of an EXEC restore that is never undone is an ordinary conditional assignment at the end of a then-block.)"""
    import re
    lines = open(asm_path).read().split("\n")
    # kernel bodies, for the "undone later" test
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)] + [len(lines)]
    found = []
    for ki in range(len(starts) - 1):
        k0, k1 = starts[ki], starts[ki + 1]
        kernel = lines[k0].rstrip(":")
        body = [l.strip() for l in lines[k0:k1]]
        copies = set(x for x in body if x.startswith("v_mov_b32_e32 v") or x.startswith("v_mov_b64_e32 v["))
        # labels reached by the branch that SKIPS a then-body: real join blocks (a label entered by s_cbranch_execnz is
        # a then-body, where a copy in front of the closing EXEC restore is an ordinary conditional assignment)
        joins = set(m.group(1) for x in body for m in [re.match(r"s_cbranch_execz (\.LBB\w+)", x)] if m)
        for i in range(k0, k1):
            if not lines[i].startswith(".LBB"):
                continue
            seen, j = [], i + 1
            while j < k1 and not lines[j].startswith(".LBB") and j < i + 14:     # an EXEC restore sits at the top of its block
                t = lines[j].strip()
                if re.match(r"s_or_b64 exec, (exec, s\[|s\[\d+:\d+\], exec)", t):
                    for x in seen:
                        m = re.match(r"(v_mov_b32_e32|v_mov_b64_e32) (v\d+|v\[\d+:\d+\]), (v\d+|v\[\d+:\d+\])$", x)
                        undone = m is not None and f"{m.group(1)} {m.group(3)}, {m.group(2)}" in copies
                        ma = re.match(r"v_accvgpr_write_b32 (a\d+), (v\d+)", x)      # AGPRs as spill space: same hazard
                        if ma is not None and any(y.startswith("v_accvgpr_read_b32") and y.endswith(", " + ma.group(1)) for y in body):
                            undone = True
                        if m is not None and lines[i].split(":")[0] in joins:
                            undone = True          # any VGPR copy at the top of a join block, before EXEC is whole again
                        if undone or x.startswith("scratch_store"):
                            found.append(f"{os.path.basename(asm_path)}:{i + 1}: {kernel[:80]}: `{x}` ahead of the EXEC restore")
                    break
                if t and not t.startswith(";"):
                    seen.append(t)
                j += 1
    return found


def _deps(name: str, seen: set | None = None) -> set:
    """The file and every header it includes with quotes, transitively (paths relative to csrc/)."""
    import re
    seen = set() if seen is None else seen
    path = os.path.normpath(os.path.join(CSRC, name))
    if path in seen:
        return seen
    seen.add(path)
    with open(path, "r", errors="replace") as f:
        for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', f.read(), re.M):
            _deps(os.path.join(os.path.dirname(name), m.group(1)), seen)
    return seen


def _unit_hash(source: str) -> str:
    """What one translation unit is built from: flags, the source, the headers it reaches."""
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for path in sorted(_deps(source)):
        with open(path, "rb") as f:
            h.update(os.path.relpath(path, CSRC).encode())
            h.update(f.read())
    return h.hexdigest()


def _build_locked(verbose: bool, force: bool = False) -> str:
    import glob
    import shutil
    import tempfile
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # the hash that will vouch for this build is the one of the sources as they are NOW: a file edited while the
    # compilers run must leave a library that needs_build() sends back here
    source_hash = _source_hash()
    objs = []
    procs = []
    tmp = tempfile.mkdtemp(prefix="molar_hip_build_")      # compiler temporaries (the device ISA is audited below)
    for s in SOURCES:
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        # an object whose unit hash (flags + source + reached headers) is unchanged was compiled AND audited before
        uh = _unit_hash(s)
        try:
            fresh = not force and os.path.exists(obj) and open(obj + ".hash").read().strip() == uh
        except OSError:
            fresh = False
        if fresh:
            continue
        if os.path.exists(obj + ".hash"):
            os.remove(obj + ".hash")
        cmd = [hipcc, *FLAGS, "-save-temps", "-c", os.path.join(CSRC, s), "-o", obj]      # temporaries go to the cwd
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, obj, uh, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=tmp)))
    failed = []
    for cmd, obj, uh, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append(f"hipcc failed: {' '.join(cmd)}\n{out}")
        elif verbose and out.strip():
            print(out)
    if failed:
        shutil.rmtree(tmp, ignore_errors=True)
        raise RuntimeError("\n".join(failed))
    hazards = []
    asms = sorted(glob.glob(os.path.join(tmp, "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    if len(asms) != len(procs):
        raise RuntimeError(f"ISA audit: expected {len(procs)} device assembly files in {tmp}, found {len(asms)}")
    for asm in asms:
        hazards += exec_restore_hazards(asm)
    shutil.rmtree(tmp, ignore_errors=True)
    if hazards and not os.environ.get("MOLAR_HIP_ALLOW_EXEC_HAZARD"):
        raise RuntimeError("the compiler placed VGPR copies ahead of an EXEC restore (see exec_restore_hazards):\n" + "\n".join(hazards))
    for cmd, obj, uh, p in procs:
        with open(obj + ".hash", "w") as f:
            f.write(uh)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed: {' '.join(cmd)}\n{r.stdout}{r.stderr}")
    with open(HASH_FILE, "w") as f:
        f.write(source_hash)
    return LIB


if __name__ == "__main__":
    import sys
    # `python -m molar_amd.build` rebuilds the units whose sources changed; `--all` recompiles every unit
    print(build_library(force="--all" in sys.argv, verbose=True))
