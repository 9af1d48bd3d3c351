"""GRO structure files for the engine's host mirror — the plumbing of BASELINE config 1 (a 25k-atom .gro box through
`within` search + fit + rmsd).  Follows MolAR's reader/writer, not the GROMACS manual where they differ:

    read     molar/src/io/gro_handler.rs:55-188   fixed columns resid[0:5] resname[5:10] name[10:15] x[20:28] y[28:36]
                                                  z[36:44] (vx,vy,vz [44:68] when the first atom line is >= 68 bytes);
                                                  time from the last "t=" of the title; box line of 3 or 9 numbers:
                                                  xx yy zz [xy xz yx yz zx zy] -> matrix with COLUMNS a, b, c (:154-185)
    write    gro_handler.rs:219-288               "{:>5.5}{:<5.5}{:>5.5}{:>5.5}{:>8.3}{:>8.3}{:>8.3}", index and
                                                  resid modulo 99999, off-diagonals only for triclinic boxes
    element  molar/src/atom.rs:238-291            SOD/POT specials; two-letter match (C,N,O,H,P-initial elements only
                                                  when name == resname, e.g. the ions CA, CL); then one-letter match;
                                                  mass / vdW from the periodic table (periodic_table.rs), vdW in nm (*0.1)
"""
from __future__ import annotations

import numpy as np

from . import api

ELEMENT_NAME = ("X H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y Zr "
                "Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir Pt "
                "Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr Rf Db Sg Bh Hs Mt Ds Rg").split()
ELEMENT_MASS = np.array([
    0.00000, 1.00794, 4.00260, 6.941, 9.012182, 10.811, 12.0107, 14.0067, 15.9994, 18.9984032, 20.1797, 22.989770, 24.3050,
    26.981538, 28.0855, 30.973761, 32.065, 35.453, 39.948, 39.0983, 40.078, 44.955910, 47.867, 50.9415, 51.9961, 54.938049,
    55.845, 58.9332, 58.6934, 63.546, 65.409, 69.723, 72.64, 74.92160, 78.96, 79.904, 83.798, 85.4678, 87.62, 88.90585, 91.224,
    92.90638, 95.94, 98.0, 101.07, 102.90550, 106.42, 107.8682, 112.411, 114.818, 118.710, 121.760, 127.60, 126.90447, 131.293,
    132.90545, 137.327, 138.9055, 140.116, 140.90765, 144.24, 145.0, 150.36, 151.964, 157.25, 158.92534, 162.500, 164.93032,
    167.259, 168.93421, 173.04, 174.967, 178.49, 180.9479, 183.84, 186.207, 190.23, 192.217, 195.078, 196.96655, 200.59, 204.3833,
    207.2, 208.98038, 209.0, 210.0, 222.0, 223.0, 226.0, 227.0, 232.0381, 231.03588, 238.02891, 237.0, 244.0, 243.0, 247.0, 247.0,
    251.0, 252.0, 257.0, 258.0, 259.0, 262.0, 261.0, 262.0, 266.0, 264.0, 269.0, 268.0, 271.0, 272.0], np.float32)
ELEMENT_VDW = np.array([
    1.5, 1.2, 1.4, 1.82, 2.0, 2.0, 1.7, 1.55, 1.52, 1.47, 1.54, 1.36, 1.18, 2.0, 2.1, 1.8, 1.8, 2.27, 1.88, 1.76, 1.37, 2.0,
    2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 1.63, 1.4, 1.39, 1.07, 2.0, 1.85, 1.9, 1.85, 2.02, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0,
    1.63, 1.72, 1.58, 1.93, 2.17, 2.0, 2.06, 1.98, 2.16, 2.1, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0,
    2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 1.72, 1.66, 1.55, 1.96, 2.02, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 1.86, 2.0,
    2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0], np.float32)
assert len(ELEMENT_NAME) == len(ELEMENT_MASS) == len(ELEMENT_VDW) == 112
_UPPER = [e.upper() for e in ELEMENT_NAME]


def guess_element(name: str, resname: str = "") -> int:
    """Atom::guess_element_from_name (atom.rs:238-283)."""
    an = 0
    i = next((k for k, ch in enumerate(name) if ch.isascii() and ch.isalpha()), None)
    if i is None:
        return 0
    if name == "SOD":
        an = 11
    elif name == "POT":
        an = 19
    if an == 0 and i + 1 < len(name):
        c2 = name[i:i + 2].upper()
        for z in range(1, len(_UPPER)):
            el = _UPPER[z]
            if len(el) == 2 and el == c2:
                if el[0] in "CNOHP":
                    if name == resname:
                        an = z
                else:
                    an = z
    if an == 0:
        for z in range(1, len(ELEMENT_NAME)):
            if len(ELEMENT_NAME[z]) == 1 and ELEMENT_NAME[z] == name[i]:
                an = z
    return an


class GroTopology(api.Topology):
    """Per-atom columns of a GRO file plus what MolAR guesses from them."""

    def __init__(self, names, resnames, resids):
        self.names, self.resnames = list(names), list(resnames)
        self.resids = np.asarray(resids, np.int32)
        cache = {}
        z = np.empty(len(self.names), np.int32)
        for k, key in enumerate(zip(self.names, self.resnames)):
            if key not in cache:
                cache[key] = guess_element(*key)
            z[k] = cache[key]
        self.atomic_numbers = z
        super().__init__(ELEMENT_MASS[z], ELEMENT_VDW[z] * np.float32(0.1))


def read_gro(path):
    """Returns (GroTopology, State).  Velocities are parsed away like the reference does when not asked for."""
    with open(path, "r") as f:
        title = f.readline()
        k = title.rfind("t=")
        time = 0.0
        if k >= 0:
            try:
                time = float(title[k + 2:].strip())
            except ValueError:
                time = 0.0
        natoms = int(f.readline().strip())
        names, resnames = [], []
        resids = np.empty(natoms, np.int32)
        xyz = np.empty((natoms, 3), np.float32)
        for a in range(natoms):
            line = f.readline()
            if len(line) < 44:
                raise ValueError(f"atom entry {a}: line too short")
            resids[a] = int(line[0:5])
            resnames.append(line[5:10].strip())
            names.append(line[10:15].strip())
            xyz[a, 0] = np.float32(line[20:28]); xyz[a, 1] = np.float32(line[28:36]); xyz[a, 2] = np.float32(line[36:44])
        l = [np.float32(x) for x in f.readline().split()]
        m = np.zeros((3, 3), np.float32)
        m[0, 0], m[1, 1], m[2, 2] = l[0], l[1], l[2]
        if len(l) == 9:
            m[1, 0], m[2, 0], m[0, 1], m[2, 1], m[0, 2], m[1, 2] = l[3], l[4], l[5], l[6], l[7], l[8]
    return GroTopology(names, resnames, resids), api.State(xyz, api.PeriodicBox.from_matrix(m), float(np.float32(time)))


def write_gro(path, top: GroTopology, state: api.State):
    m = state.pbox.get_matrix() if state.pbox is not None else None
    with open(path, "w") as f:
        f.write(f"Created by Molar, t= {state.time:.3f}\n{len(state)}\n")
        for i in range(len(state)):
            p = state.coords[i]
            f.write("%5.5s%-5.5s%5.5s%5.5s%8.3f%8.3f%8.3f\n" % (str(int(top.resids[i]) % 99999), top.resnames[i], top.names[i],
                                                              str(i % 99999 + 1), p[0], p[1], p[2]))
        if m is None:
            f.write("0.0 0.0 0.0\n")
        else:
            f.write("%10.4f %10.4f %10.4f" % (m[0, 0], m[1, 1], m[2, 2]))
            if state.pbox.is_triclinic():
                f.write(" %10.4f %10.4f %10.4f %10.4f %10.4f %10.4f" % (m[1, 0], m[2, 0], m[0, 1], m[2, 1], m[0, 2], m[1, 2]))
            f.write("\n")
