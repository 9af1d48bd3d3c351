"""CPU columns of the tools' benchmarks: the oracle timed at 1, 8 and all host threads; the best is what a GPU time is compared
with, the all-threads figure is kept beside it.  (Round 4's columns forked a team of every host thread over plans of a few
hundred entries: ~250 ms whatever the problem.  The oracle now sizes its team by the plan, oracle/molar_oracle.c run_plan.)"""
import os
import time


def cpu_best(fn_of_threads, reps=2, min_seconds=0.15):
    """fn_of_threads(nthreads) -> result.  Returns (best_seconds, result, {"cpu_threads_best", "cpu_ms_by_threads", "host_cores"})."""
    ncores = os.cpu_count() or 1
    by, out = {}, None
    for nt in sorted({1, min(8, ncores), ncores}):
        fn_of_threads(nt)                      # warm-up (thread team, allocator)
        t0 = time.perf_counter()
        k = 0
        while k < reps or time.perf_counter() - t0 < min_seconds:
            out = fn_of_threads(nt)
            k += 1
        by[nt] = (time.perf_counter() - t0) / k
    best = min(by, key=by.get)
    return by[best], out, {"cpu_threads_best": best, "cpu_ms_by_threads": {str(k): v * 1e3 for k, v in by.items()}, "host_cores": ncores}
