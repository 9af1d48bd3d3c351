"""Deterministic synthetic frames for the benchmark configs.

Positions are uniform in fractional coordinates mapped through the box matrix, then
jittered with a Gaussian (sigma nm) so that a few percent of atoms leave the primary
cell and exercise the wrap branch of the PBC binning (distance_search.rs:181-199).
Number density defaults to 100 atoms/nm^3 (water-like).  Pure numpy, host side.
"""
from __future__ import annotations

import numpy as np

SEED = 20240607
MASS_CYCLE = np.array([1.008, 12.011, 14.007, 15.999], dtype=np.float32)


def box_a(natoms: int, density: float = 100.0) -> np.ndarray:
    """'Box A' (benign triclinic, negative shear): columns a,b,c; reference grid complete.

    For natoms=1e6 this is a=(21.544,0,0), b=(0,21.544,0), c=(-3,-3,21.544) of SURVEY §8d;
    other sizes scale every length by (natoms/1e6)^(1/3).
    """
    s = (natoms / density / 1.0e4) ** (1.0 / 3.0)
    L = 21.544 * s
    sh = -3.0 * s
    return np.array([[L, 0.0, sh], [0.0, L, sh], [0.0, 0.0, L]], dtype=np.float32)


def box_b(natoms: int, density: float = 100.0) -> np.ndarray:
    """'Box B' (GROMACS-style hexagonal prism): a=(L,0,0), b=(L/2, L*sqrt(3)/2, 0), c=(0,0,L')."""
    vol = natoms / density
    L = (vol / (np.sqrt(3.0) / 2.0)) ** (1.0 / 3.0)
    return np.array([[L, L / 2.0, 0.0], [0.0, L * np.sqrt(3.0) / 2.0, 0.0], [0.0, 0.0, L]], dtype=np.float32)


def box_ortho(natoms: int, density: float = 100.0) -> np.ndarray:
    L = (natoms / density) ** (1.0 / 3.0)
    return np.diag([L, L, L]).astype(np.float32)


def frame(natoms: int, box: np.ndarray, frame_no: int = 0, sigma: float = 0.05, seed: int = SEED) -> np.ndarray:
    """One frame: natoms x 3 float32 (AoS, MolAR's State.coords layout, state.rs:22-28)."""
    base = np.random.default_rng(seed)
    frac = base.random((natoms, 3), dtype=np.float64)
    pos = frac @ box.astype(np.float64).T
    jit = np.random.default_rng(seed + 1 + frame_no).normal(0.0, sigma, size=(natoms, 3))
    return (pos + jit).astype(np.float32)


def masses(natoms: int) -> np.ndarray:
    return np.resize(MASS_CYCLE, natoms).astype(np.float32)
