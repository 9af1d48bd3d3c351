#!/usr/bin/env python
"""gpurun_out/r06 (raw output of tools/r06_final.sh) -> the committed files of profiles/ (r06_*): tools/refresh_profiles.py for the headline
set, plain copies for the rest, the cutoff-sweep table from its JSON lines (round 5's column from profiles/r05_cutoff_sweep.jsonl).
usage: python tools/r06_collect.py"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out", "r06"), os.path.join(ROOT, "profiles")


def last_json(path):
    return [l for l in open(path).read().splitlines() if l.startswith("{")]


def main():
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "refresh_profiles.py")], env=dict(os.environ, ROUND="r06"), check=True)
    copies = {"bench_steps20.json": "r06_bench_steps20.json", "rdf_bench_single_calls.json": "r06_rdf_bench_single_calls.json",
              "rdf_xtc.jsonl": "r06_rdf_xtc.jsonl", "membrane_bench.json": "r06_membrane_bench.json", "bench_configs.jsonl": "r06_bench_configs.jsonl",
              "frame_timeline.txt": "r06_frame_timeline.txt", "sq_pair.txt": "r06_sq_instruction_mix.txt", "sq_hist.txt": "r06_hist_sq_mix.txt",
              "final_rdf_trace.txt": "r06_rdf_frame_trace.txt", "final_membrane_kernels.csv": "r06_membrane_kernels.csv",
              "cutoff_sweep.jsonl": "r06_cutoff_sweep.jsonl", "final_bench_driver.json": "r06_bench_driver_style.json",
              "final_c5_timeline.txt": "r06_membrane_timeline.txt", "final_tests.txt": "r06_gpu_tests.txt"}
    for src, dst in copies.items():
        s = os.path.join(G, src)
        if not os.path.exists(s):
            print("missing", src)
            continue
        if src.endswith((".json", ".jsonl")):
            open(os.path.join(P, dst), "w").write("\n".join(last_json(s)) + "\n")
        elif src in ("sq_pair.txt", "sq_hist.txt"):
            open(os.path.join(P, dst), "w").write("".join(l for l in open(s) if not l.startswith("/")))
        else:
            shutil.copy(s, os.path.join(P, dst))
    # the sweep's table
    old = {}
    for l in last_json(os.path.join(P, "r05_cutoff_sweep.jsonl")):
        d = json.loads(l)
        old[round(d["cutoff_nm"], 3)] = d.get("mpairs_per_ms_pipelined")
    rows = [json.loads(l) for l in last_json(os.path.join(P, "r06_cutoff_sweep.jsonl"))]

    def kernels(d):
        apc = d["atoms_per_cell"]
        if apc <= 13: return "small-cell kernels, 16 lanes per slot (pair_small.hip)"
        if apc <= 19: return "small-cell kernels, 32 lanes per slot"
        if apc <= 448: return "regular (matrix-core count up to 320 atoms per second cell)"
        if apc <= 1000: return "128-register instances, <= 1024 atoms resident (pair_k5/6.hip)"
        return "168-register instances, <= 2048 atoms resident (pair_k7/8.hip, round 6)" + (" - most cells above 2048: streamed" if apc > 2048 else "")

    with open(os.path.join(P, "r06_cutoff_sweep.txt"), "w") as f:
        f.write("# Ordered pair list in HBM for the 1M-atom frame of the headline (triclinic box A, 100 atoms / nm^3; distance_search_single_pbc) over the\n"
                "# cutoff, one MI355X, end of round 6 (tools/bench_cutoff_sweep.py, raw lines: r06_cutoff_sweep.jsonl).  pipelined = _resident_begin / _end with\n"
                "# two frames in flight; count / fill = HIP-event brackets of the two passes in a separate profiled pass of the same loop (5-10 us each above the\n"
                "# kernels); roofline = (12 N + 12 P) bytes / pipelined time / 8 TB/s (SURVEY.md 8d); r05 = the same column at the end of round 5.\n#\n"
                "#  rc / nm  atoms/cell     pairs   pipelined / ms  count / ms  fill / ms  M pairs per ms   r05   HBM roofline   kernels\n")
        for d in rows:
            k = d["kernel_ms_per_frame"]
            o = old.get(round(d["cutoff_nm"], 3))
            f.write(f"#  {d['cutoff_nm']:<7g} {d['atoms_per_cell']:8.1f} {d['pairs']:10.3g} {d['ms_resident_pipelined']:13.3f} {k['pair_count']:11.3f} "
                    f"{k['pair_fill']:10.3f} {d['mpairs_per_ms_pipelined']:14.1f} {(f'{o:.1f}' if o else '-'):>7s} {d['hbm_roofline_frac_pipelined']:11.3f}     {kernels(d)}\n")
    print(open(os.path.join(P, "r06_cutoff_sweep.txt")).read())
    for name in ("r06_membrane_bench.json", "r06_rdf_xtc.jsonl", "r06_rdf_bench_single_calls.json", "r06_bench_driver_style.json"):
        for l in last_json(os.path.join(P, name)):
            d = json.loads(l)
            print(name, round(d["value"], 1), d.get("steps"), [(s.get("config", {}).get("workload", "?")[:24], round(s["value"], 1)) for s in d.get("secondary", []) if isinstance(s, dict)])


if __name__ == "__main__":
    main()
