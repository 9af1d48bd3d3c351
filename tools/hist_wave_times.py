"""Per-wave time accounting of hist_kernel on the C4 frame (250k atoms, rc 1.2, 1200 bins): where the persistent waves spend
their time and how far apart they finish.  Needs a library built with -DMOLAR_HIP_DEBUG_KNOBS (tools/build_variant.sh dbg ...):

    MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_dbg.so python tools/hist_wave_times.py

The kernel writes 8 words per wave (s_memrealtime, 100 MHz): start, end, time between taking a slot and entering its row loop,
time inside row loops (incl. draining the stack), the wrapped entries' share of that, slots by class, XCC id."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molar_amd import api, synth  # noqa: E402

n, nbins, rc = 250_000, 1200, 1.2
box = synth.box_a(n)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(20240607)
base = torch.rand((n, 3), generator=g, device=dev, dtype=torch.float64) @ torch.from_numpy(box.astype(np.float64)).to(dev).T
frame = (base + (torch.randn((n, 3), generator=g, device=dev, dtype=torch.float32) * 0.05).double()).float()
# MOLAR_HIP_DEBUG_LAUNCH=n records the n-th launch only: with 12 launches queued, n = 8 sits in the steady state of the pipeline
NL = 12 if os.environ.get("MOLAR_HIP_DEBUG_LAUNCH") else 5
eng = api.Engine(0)
bins = torch.zeros(nbins, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for _ in range(NL):
    eng.search_histogram(api.SEARCH_SINGLE, rc, 0.0, rc, nbins, frame, box=box, pbc=7, bins=bins, want_count=False)
eng.synchronize()
nw = 256 * 32
buf = np.zeros((nw, 16), np.uint64)
fetch = eng.lib.molar_hip_debug_fetch
fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
rc_ = fetch(eng.ctx, buf.ctypes.data, buf.nbytes)
assert rc_ == 0, rc_
t0, t1 = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64)
ok = t1 > 0
t0, t1, b = t0[ok], t1[ok], buf[ok]
start, end = t0.min(), t1.max()
span = (end - start) * 0.01          # us
life = (t1 - t0) * 0.01
pre, rows, wr = b[:, 2] * 0.01, b[:, 3] * 0.01, b[:, 3] * 0.0
nplain, nwrap, ntri = np.ones(len(b), np.int64), np.zeros(len(b), np.int64), np.zeros(len(b), np.int64)
idle_tail = (end - t1) * 0.01
late = (t0 - start) * 0.01
out = {
    "waves": int(ok.sum()), "kernel_span_us": round(span, 1),
    "wave_life_us_mean": round(float(life.mean()), 1),
    "start_delay_us": {"mean": round(float(late.mean()), 1), "p50": round(float(np.percentile(late, 50)), 1), "p90": round(float(np.percentile(late, 90)), 1), "max": round(float(late.max()), 1)},
    "idle_after_last_slot_us": {"mean": round(float(idle_tail.mean()), 1), "p50": round(float(np.percentile(idle_tail, 50)), 1), "p90": round(float(np.percentile(idle_tail, 90)), 1), "max": round(float(idle_tail.max()), 1)},
    "share_of_span": {"before_start": round(float(late.mean() / span), 3), "slot_preamble": round(float(pre.mean() / span), 3),
                      "row_loops": round(float(rows.mean() / span), 3), "of_which_wrapped": round(float(wr.mean() / span), 3),
                      "idle_at_end": round(float(idle_tail.mean() / span), 3)},
    "slots_per_wave": {"plain": round(float(nplain.mean()), 2), "wrapped": round(float(nwrap.mean()), 2), "same_cell": round(float(ntri.mean()), 2)},
    "us_per_slot": {"plain+same_cell": round(float((rows.sum() - wr.sum()) / max((nplain + ntri).sum(), 1)), 1), "wrapped": round(float(wr.sum() / max(nwrap.sum(), 1)), 1),
                    "preamble": round(float(pre.sum() / max((nplain + ntri + nwrap).sum(), 1)), 1)},
}
mx, last = b[:, 8] * 0.01, (b[:, 9].astype(np.int64) - start) * 0.01
out["longest_slot_us"] = {"p50": round(float(np.percentile(mx, 50)), 1), "p90": round(float(np.percentile(mx, 90)), 1), "p99": round(float(np.percentile(mx, 99)), 1), "max": round(float(mx.max()), 1)}
out["last_slot_start_us"] = {"p10": round(float(np.percentile(last, 10)), 1), "p50": round(float(np.percentile(last, 50)), 1), "p90": round(float(np.percentile(last, 90)), 1), "max": round(float(last.max()), 1)}
# the ten longest slots: flags (wrap | tri<<8 | ...), chunks, hits, duration, when they started
top = np.argsort(-mx)[:24]
out["longest_slots"] = [{"us": round(float(mx[i]), 1), "slot": int(b[i, 10] >> 32), "flags": hex(int(b[i, 10] & 0xFFFF)), "nch": int((b[i, 10] >> 16) & 15),
                         "hits": int((b[i, 10] >> 20) & 0xFFF), "wg": int(i // 16), "wave": int(i % 16), "xcc": int(b[i, 7] & 0xF), "ticket_wait_us": round(float(b[i, 12] * 0.01), 1), "preamble_us": round(float(b[i, 11] * 0.01), 1), "start_us": round(float((int(b[i, 13]) - start) * 0.01), 1), "wave_end_us": round(float((t1[i] - start) * 0.01), 1)} for i in top]
def _cls(col):
    t = float((b[:, col] & ((1 << 40) - 1)).sum()) * 0.01
    k = int((b[:, col] >> 40).sum())
    return {"slots": k, "us_per_slot": round(t / max(k, 1), 1)}
out["by_class"] = {"same_cell_5_chunks": _cls(15), "same_cell_6_chunks": _cls(4), "plain_5_chunks": _cls(5), "plain_6_chunks": _cls(6)}
first = b[:, 14] * 0.01
out["first_slot_us"] = {"p50": round(float(np.percentile(first, 50)), 1), "p90": round(float(np.percentile(first, 90)), 1), "max": round(float(first.max()), 1)}
st = (b[:, 13].astype(np.int64) - start) * 0.01
out["longest_slot_start_us"] = {"p10": round(float(np.percentile(st, 10)), 1), "p50": round(float(np.percentile(st, 50)), 1), "p90": round(float(np.percentile(st, 90)), 1)}
out["executed_steps"] = int((b[:, 7] >> 8).sum())          # (row, 64-atom chunk) steps the lean kernel evaluated in this launch
out["executed_candidate_evaluations"] = out["executed_steps"] * 64
xcc = (b[:, 7] & 0xF).astype(np.int64)
out["end_by_xcc_us"] = {int(x): round(float(((t1[xcc == x]).max() - start) * 0.01), 1) for x in np.unique(xcc)}
# per workgroup (16 consecutive waves): the last wave's end
wg_end = (t1.reshape(-1, 16).max(axis=1) - start) * 0.01 if len(t1) % 16 == 0 else None
if wg_end is not None:
    out["workgroup_end_us"] = {"min": round(float(wg_end.min()), 1), "p10": round(float(np.percentile(wg_end, 10)), 1), "p50": round(float(np.percentile(wg_end, 50)), 1),
                               "p90": round(float(np.percentile(wg_end, 90)), 1), "max": round(float(wg_end.max()), 1)}
print(json.dumps(out))
