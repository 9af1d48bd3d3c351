#!/usr/bin/env python
"""Device-side XTC decode (molar_hip_xtc_read_device: one lane per frame) against the host decoder threads, 250k-atom frames:
frames/s by window size.  The file is synthetic (four distinct frames cycled, written with the test encoder of oracle/ -
measurement infrastructure, like tools/rdf_xtc.py)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from molar_amd import api, build, synth
    from molar_amd.xtc import XtcReader
    from oracle.oracle import Oracle
    build.build_library()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    o = Oracle("f32")
    box = synth.box_a(n)
    blobs = [o.xtc_encode(synth.frame(n, box, f), np.ascontiguousarray(box.T).reshape(9), step=f, time=float(f)) for f in range(4)]
    data = b"".join(blobs[k % 4] for k in range(F))
    eng = api.Engine(0)
    r = XtcReader(data, engine=eng, nthreads=os.cpu_count())
    out = torch.empty((F, n, 3), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ref = r.read_frames(0, 4)
    for W in (64, 256, 1024, 2048):
        if W > F: continue
        r.read_frames_device(0, W, out[:W])
        t0 = time.perf_counter()
        r.read_frames_device(0, W, out[:W])
        dt = time.perf_counter() - t0
        ok = bool(np.array_equal(out[:4].cpu().numpy(), ref))
        print(json.dumps({"decoder": "device, one lane per frame", "natoms": n, "window_frames": W, "seconds": dt, "frames_per_s": W / dt,
                          "compressed_MB_per_frame": len(blobs[0]) / 1e6, "bits_equal_host": ok}), flush=True)
    for T in (16, 64, os.cpu_count()):
        W = min(F, 512)
        r.read_frames(0, W, out=out[:W], nthreads=T)
        t0 = time.perf_counter()
        r.read_frames(0, W, out=out[:W], nthreads=T)
        dt = time.perf_counter() - t0
        print(json.dumps({"decoder": f"host, {T} threads -> HBM", "natoms": n, "window_frames": W, "seconds": dt, "frames_per_s": W / dt,
                          "frames_per_s_per_thread": W / dt / T}), flush=True)


if __name__ == "__main__":
    main()
