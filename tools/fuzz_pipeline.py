#!/usr/bin/env python
"""Randomised test of the pipelined resident search (molar_hip_search_resident_begin/_end): random frames of random
size, box, cutoff, kind and periodicity go through begin/end with two searches in flight (ended in order or out of
order, now and then with a plain call in between), and every result must equal count + fill of the same frame on a
second context - the path the oracle fuzz (tools/fuzz_search.py) checks.  Exercises the two grid generations, the side
stream of the grid build and the grow-and-repeat logic.
Usage: python tools/fuzz_pipeline.py [nframes] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Dev:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def run(nframes=300, seed=1, verbose=True):
    import torch
    from molar_amd import api, build
    from tools.fuzz_search import random_box
    build.build_library()
    ref_eng, eng = api.Engine(0), api.Engine(0)
    rng = np.random.default_rng(seed)
    fails = 0
    inflight = []      # (ticket, want, keepalive)

    def fetch(res):
        cnt, pp, dp = res
        if cnt == 0:
            return np.zeros((0, 2), np.uint32), np.zeros(0, np.float32)
        p = torch.as_tensor(_Dev(pp, cnt * 2, "<i4"), device="cuda").cpu().numpy().view(np.uint32).reshape(-1, 2)
        d = torch.as_tensor(_Dev(dp, cnt, "<f4"), device="cuda").cpu().numpy()
        return p, d

    def check(tag, res, want):
        nonlocal fails
        p, d = fetch(res)
        if not (res[0] == want[0] and np.array_equal(p, want[1]) and np.array_equal(d, want[2])):
            fails += 1
            print("MISMATCH", tag, res[0], want[0])
            if res[0] == want[0] and os.environ.get("MOLAR_FUZZ_VERBOSE"):
                bp = np.nonzero((p != want[1]).any(axis=1))[0]
                bd = np.nonzero(d != want[2])[0]
                print("  pair rows differing:", len(bp), bp[:8], "dist differing:", len(bd), bd[:8])
                if len(bp):
                    print("  got", p[bp[:4]].tolist(), "want", want[1][bp[:4]].tolist())
                if len(bd):
                    print("  got", d[bd[:4]].tolist(), "want", want[2][bd[:4]].tolist())

    def end_one(k):
        t, want, keep, tag = inflight.pop(k)
        check(tag, eng.search_resident_end(t), want)

    for f in range(nframes):
        box = random_box(rng)
        vol = abs(np.linalg.det(box.astype(np.float64)))
        n = int(min(max(vol * rng.choice([20.0, 60.0, 100.0, 100.0, 600.0, 2000.0]), 50), rng.choice([3000, 12000, 40000])))
        pos_h = (rng.random((n, 3)) @ box.astype(np.float64).T + rng.normal(0, rng.choice([0.0, 0.05, 0.5]), (n, 3))).astype(np.float32)
        pos = torch.from_numpy(pos_h).cuda()
        rc = float(np.float32(rng.uniform(0.25, 1.0)))
        pbc = int(rng.choice([7, 7, 7, 7, 3, 5, 0]))
        kw = dict(box=box, pbc=pbc) if pbc else {}
        kind = int(rng.choice([api.SEARCH_SINGLE, api.SEARCH_SINGLE, api.SEARCH_DOUBLE]))
        try:
            if kind == api.SEARCH_SINGLE:
                idx = None if rng.random() < 0.6 else torch.from_numpy(np.sort(rng.choice(n, max(n // 2, 2), replace=False)).astype(np.int64)).cuda()
                args, dkw = (kind, rc, pos, idx), dict(idx1=idx)
            else:
                perm = rng.permutation(n)
                i1 = torch.from_numpy(np.sort(perm[: n // 3]).astype(np.int64)).cuda()
                i2 = torch.from_numpy(np.sort(perm[n // 3:]).astype(np.int64)).cuda()
                args, dkw = (kind, rc, pos, i1, pos, i2), dict(idx1=i1, xyz2=pos, idx2=i2)
            wn = ref_eng.search_count(*args, **kw)
            want = (wn,) + tuple(ref_eng.search_fill(wn))
            tag = f"frame {f}: kind {kind} n {n} rc {rc:.3f} pbc {pbc}"
            if rng.random() < 0.1:                      # a plain call between the pipelined ones (it shares result set 0,
                while inflight:                         # so the tickets are ended first)
                    end_one(0)
                check(tag + " (plain)", eng.search_resident(*args, **kw), want)
                continue
            if len(inflight) == 2:
                if rng.random() < 0.3:
                    end_one(1)                          # out of order: the younger search first ...
                end_one(0)                              # ... the next begin reuses the older one's result set
            desc, keep = eng.make_search_desc(kind, rc, pos, **dkw, **kw)
            t = eng.search_resident_begin(desc)
            inflight.append((t, want, (desc, keep, pos, args), tag))
            if rng.random() < 0.2:                      # sometimes drain completely
                while inflight:
                    end_one(0)
        except Exception as exc:
            fails += 1
            print("ERROR", f, repr(exc))
            inflight.clear()
            eng = api.Engine(0)
    while inflight:
        end_one(0)
    if verbose:
        print(f"{nframes} frames, {fails} failures")
    return fails


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
