"""bench.py's launch path without a GPU: `--gpus N` from a bare shell must spawn N ranks by itself (gloo here), run the
end-of-run reductions and print ONE JSON line on rank 0; on a node with too few GPUs it must refuse with a message, not
a traceback; and a run that was launched with the wrong world size must say so."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=280, env=e, cwd=ROOT)


@pytest.mark.timeout(300)
def test_self_launch_two_ranks_gloo():
    r = run("--gpus", "2", "--steps", "7", "--launch-check", "--backend", "gloo")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                     # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["frames_per_gpu"] == 7
    assert d["pairs"] == 3 and d["bins_sum"] == 1200 * 3 and d["max_over_ranks"] == 2.0     # the all_reduces really ran


@pytest.mark.timeout(120)
def test_refuses_cleanly_without_enough_gpus():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("node has 2 GPUs")
    r = run("--gpus", "2")
    assert r.returncode != 0 and "Traceback" not in r.stderr
    assert "--gpus 2" in r.stderr and "GPU(s)" in r.stderr


@pytest.mark.timeout(120)
def test_world_size_mismatch_is_reported():
    r = run("--gpus", "2", "--launch-check", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.timeout(300)
@pytest.mark.parametrize("gpus", [1, 2])
def test_xtc_frame_supply_of_the_rdf_workload(gpus):
    """`--workload rdf --source xtc` without the GPU: rank 0 writes the synthetic trajectory with the library's XTC writer,
    every rank decodes its own contiguous block on host threads and gets exactly the frames (on the format's grid); the
    ranks' counts are reduced over gloo."""
    r = run("--gpus", str(gpus), "--steps", "5", "--launch-check", "--backend", "gloo", "--workload", "rdf", "--source", "xtc")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["source"] == "xtc" and d["n_gpus"] == gpus
    assert d["frames_in_file"] == 5 * gpus and d["frames_decoded"] == 5 * gpus and d["ranks_with_exact_frames"] == gpus
