"""Where do a kernel's register spills execute?  Reads the device assembly of one translation unit (hipcc -save-temps: the
*-hip-amdgcn-amd-amdhsa-gfx950.s file) and, per kernel, lists every scratch_load / scratch_store and every v_writelane /
v_readlane to a spill VGPR with the innermost loop that contains it (a loop = a backward branch to a label; its body = the lines
between the label and the branch) and whether that loop holds matrix or candidate arithmetic (v_mfma / v_pk_mul_f32 / v_pk_fma_f32).

    python tools/analysis/spill_sites.py /tmp/isa/pair_k0-hip-amdgcn-amd-amdhsa-gfx950.s [kernel-substring]
"""
import re
import sys


def kernels(lines):
    cur, start = None, 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur, start = m.group(1), i
        elif l.startswith(".Lfunc_end") and cur:
            yield cur, start, i
            cur = None


def loops(body):
    """(first, last, label) of every backward branch"""
    at = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            at[m.group(1)] = i
    out = []
    for i, l in enumerate(body):
        m = re.search(r"\bs_c?branch\w*\s+(\.LBB\w+)", l)
        if m and m.group(1) in at and at[m.group(1)] <= i:
            out.append((at[m.group(1)], i, m.group(1)))
    return out


def main():
    lines = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2] if len(sys.argv) > 2 else "pair_kernel"
    for name, a, b in kernels(lines):
        if want not in name:
            continue
        body = lines[a:b]
        lp = loops(body)
        hot = lambda s, e: sum(1 for l in body[s:e + 1] if re.search(r"v_mfma|v_pk_mul_f32|v_pk_fma_f32|v_pk_add_f32", l))
        def innermost(i):
            best = None
            for s, e, lab in lp:
                if s <= i <= e and (best is None or e - s < best[1] - best[0]):
                    best = (s, e, lab)
            return best
        rows = {}
        total = {"scratch_load": 0, "scratch_store": 0, "v_writelane": 0, "v_readlane": 0}
        for i, l in enumerate(body):
            for op in total:
                if re.search(r"\b" + op, l):
                    total[op] += 1
                    L = innermost(i)
                    key = ("outside any loop", 0, 0) if L is None else (L[2], L[1] - L[0] + 1, hot(L[0], L[1]))
                    rows.setdefault(key, dict.fromkeys(total, 0))[op] += 1
        mf = sum(1 for l in body if "v_mfma" in l)
        print(f"{name[:70]}: {len(body)} lines, {mf} v_mfma, {len(lp)} loops; static totals {total}")
        print("  innermost loop (label, lines, arithmetic instructions in it) -> spill instructions inside it")
        for key, v in sorted(rows.items(), key=lambda kv: -kv[0][2]):
            if key[0] == "outside any loop":
                continue
            sc = v["scratch_load"] + v["scratch_store"]
            if key[2] == 0 and sc == 0:
                continue          # SGPR<->lane moves in bookkeeping loops without arithmetic: not listed one by one
            print(f"    {key[0]:<14} {key[1]:>6} lines {key[2]:>5} arith   {v}")
        o = rows.get(("outside any loop", 0, 0), {})
        print(f"    outside any loop: {o}")
        cold = dict.fromkeys(total, 0)
        for key, v in rows.items():
            if key[0] != "outside any loop" and key[2] == 0:
                for k in v:
                    cold[k] += v[k]
        print(f"    in loops without candidate arithmetic: {cold}")


if __name__ == "__main__":
    main()
