#!/usr/bin/env python
"""Runs the build's ISA audit (molar_amd.build.exec_restore_hazards) over assembly files kept with -save-temps.
Usage: python tools/scan_exec_copies.py file.s ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molar_amd.build import exec_restore_hazards

if __name__ == "__main__":
    found = [h for p in sys.argv[1:] for h in exec_restore_hazards(p)]
    print("\n".join(found))
    print("suspicious copies:", len(found))
    sys.exit(1 if found else 0)
