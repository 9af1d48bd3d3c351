"""Compiles and runs the C++ host mirror (include/molar_hip.hpp) tests with g++."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "cpp", "_build")


def _compile(src, exe, extra=()):
    from molar_amd import build
    build.build_library()
    os.makedirs(OUT, exist_ok=True)
    libdir = os.path.join(ROOT, "molar_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", src),
           "-o", os.path.join(OUT, exe), "-L", libdir, "-lmolar_hip", "-lpthread", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return os.path.join(OUT, exe)


def test_analysis_task_and_suffix_cpu():
    exe = _compile("test_analysis_task.cpp", "test_analysis_task")
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:      # (the sharded-reader test writes a 40-frame, 250k-atom trajectory there: ~40 MB)
        r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "benzene.xtc"), tmp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all host-mirror CPU tests passed" in r.stdout
    assert "own readers" in r.stdout
    print(r.stdout)


def test_rotation_solver_cpu():
    """molar_amd/csrc/linalg3.hpp on the host (hipcc, no GPU needed to run): Newton + adjugate vs Jacobi, known rotations."""
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, "test_linalg")
    cmd = ["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-std=c++17",
           os.path.join(ROOT, "tests", "cpp", "test_linalg.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all linalg tests passed" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_host_api_gpu():
    from oracle import oracle as o
    o.build()
    odir = os.path.join(ROOT, "oracle")
    exe = _compile("test_host_api_gpu.cpp", "test_host_api_gpu",
                   extra=["-I", odir, "-L", odir, "-l:liboracle_f32.so", f"-Wl,-rpath,{odir}", "-L", "/opt/rocm/lib", "-lamdhip64"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all host-mirror GPU tests passed" in r.stdout
