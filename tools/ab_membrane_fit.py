#!/usr/bin/env python
"""Bit-for-bit comparison of two builds of the library on the membrane frame: tools/ab_membrane_fit.py dump OUT.npz runs a set of
bilayer frames (sizes, patch cutoffs - long patches and large Voronoi cells included -, jitter, two smoothing iterations, undulated
sheets) through Membrane.compute of the library MOLAR_HIP_PLUGIN selects and stores every output array; `compare A.npz B.npz`
checks that the two files are identical.  Used for k_membrane_fit_lanes (16 lanes per lipid) against the one-lane kernel
(-DMH_FIT_ONE_LANE, tools/build_variant.sh)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dump(path):
    from molar_amd import api, build
    from molar_amd import membrane as mb
    build.build_library()
    eng = api.Engine(0)
    out = {}
    cases = ((200, 30_000, 2.0, 1, 0.0), (800, 120_000, 2.5, 1, 0.3), (2000, 500_000, 2.5, 1, 0.0),
             (450, 60_000, 3.5, 2, 0.5), (300, 40_000, 5.0, 1, 0.2), (128, 20_000, 1.2, 1, 0.0))
    only = os.environ.get("AB_CASES")
    for ci, (nl, natoms, cutoff, iters, amp) in enumerate(cases):
        case = ci
        if only and str(ci) not in only.split(","):
            continue
        print("case", ci, nl, natoms, cutoff, iters, amp, flush=True)
        xyz, box, first, tpl, masses = mb.build_bilayer(nl, natoms, seed=100 + case)
        rng = np.random.default_rng(7 + case)
        m = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(cutoff=cutoff, order_type=1, max_smooth_iter=iters))
        for f in range(3):
            fr = xyz + rng.normal(0, 0.03, xyz.shape).astype(np.float32)
            if amp:        # an undulated sheet: curvature, more varied cells
                L = float(box[0, 0])
                fr[:, 2] += (amp * np.sin(2 * np.pi * (fr[:, 0] / L) * (1 + f)) * np.cos(2 * np.pi * fr[:, 1] / L)).astype(np.float32)
            r = m.compute(fr.astype(np.float32), box)
            for k, v in r.items():
                if isinstance(v, np.ndarray):
                    out[f"c{case}_f{f}_{k}"] = v
                elif isinstance(v, list):
                    for q, a in enumerate(v):
                        out[f"c{case}_f{f}_{k}{q}"] = np.asarray(a)
    np.savez(path, **out)
    nv = [out[k] for k in out if k.endswith("_nvert")]
    print(f"dumped {len(out)} arrays; vertices per cell up to {max(int(a.max()) for a in nv)}; "
          f"valid lipid-frames {sum(int(out[k].sum()) for k in out if k.endswith('_valid'))}")


def compare(a, b):
    A, B = np.load(a), np.load(b)
    bad = [k for k in A.files if k not in B.files or A[k].shape != B[k].shape or A[k].tobytes() != B[k].tobytes()]
    print(f"{len(A.files)} arrays, {len(bad)} differ" + (": " + ", ".join(bad[:8]) if bad else " - bit-identical"))
    return 1 if bad or set(A.files) != set(B.files) else 0


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2])
    else:
        sys.exit(compare(sys.argv[2], sys.argv[3]))
