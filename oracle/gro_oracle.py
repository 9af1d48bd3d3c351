"""TEST INFRASTRUCTURE ONLY (see molar_oracle.h).  Restatement of MolAR's GRO reader and element/mass guessing for
the config-1 plumbing test: molar/src/io/gro_handler.rs:55-188 (fixed columns, box line order
xx yy zz xy xz yx yz zx zy -> columns a,b,c) and molar/src/atom.rs:238-291 + periodic_table.rs.
PARITY UNPINNED: the reference's GRO tests need files that are not in the mount; written from the source."""
import numpy as np

_NAMES = ("X H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y "
          "Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba").split()      # enough of the table for the tests
_MASS = {"X": 0.0, "H": 1.00794, "C": 12.0107, "N": 14.0067, "O": 15.9994, "Na": 22.989770, "Mg": 24.3050, "P": 30.973761,
         "S": 32.065, "Cl": 35.453, "K": 39.0983, "Ca": 40.078, "Fe": 55.845, "Zn": 65.409}


def element_of(name, resname):
    first = next((k for k, c in enumerate(name) if c.isascii() and c.isalpha()), None)
    if first is None:
        return "X"
    if name in ("SOD", "POT"):
        return {"SOD": "Na", "POT": "K"}[name]
    found = None
    two = name[first:first + 2].upper() if first + 1 < len(name) else None
    if two:
        for el in _NAMES[1:]:
            if len(el) == 2 and el.upper() == two and (el[0] not in "CNOHP" or name == resname):
                found = el
    if found is None:
        for el in _NAMES[1:]:
            if len(el) == 1 and el == name[first]:
                found = el
    return found or "X"


def read_gro(path):
    lines = open(path).read().split("\n")
    title = lines[0]
    t = 0.0
    if "t=" in title:
        try:
            t = float(title[title.rfind("t=") + 2:].strip())
        except ValueError:
            t = 0.0
    n = int(lines[1])
    rec = lines[2:2 + n]
    resid = np.array([int(r[:5]) for r in rec], np.int32)
    resname = [r[5:10].strip() for r in rec]
    name = [r[10:15].strip() for r in rec]
    xyz = np.array([[np.float32(r[20:28]), np.float32(r[28:36]), np.float32(r[36:44])] for r in rec], np.float32)
    b = [np.float32(x) for x in lines[2 + n].split()]
    box = np.zeros((3, 3), np.float32)
    box[0, 0], box[1, 1], box[2, 2] = b[:3]
    if len(b) == 9:
        # file order: v1(y) v1(z) v2(x) v2(z) v3(x) v3(y); matrix columns are v1, v2, v3
        box[1, 0], box[2, 0], box[0, 1], box[2, 1], box[0, 2], box[1, 2] = b[3:]
    mass = np.array([_MASS.get(element_of(a, r), np.nan) for a, r in zip(name, resname)], np.float32)
    return dict(time=np.float32(t), resid=resid, resname=resname, name=name, xyz=xyz, box=box, mass=mass)
