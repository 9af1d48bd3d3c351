"""The Rust shim (rust/molar_hip) cannot be compiled here (no toolchain): this is the substitute.  It parses
include/molar_hip.h and the crate's sources and checks that they describe the same ABI."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_ffi as gen  # noqa: E402

CRATE = os.path.join(ROOT, "rust", "molar_hip")


def header_functions():
    return gen.c_functions(open(gen.HEADER).read())


def split_args(text):
    """top-level comma split of a Rust argument list"""
    out, depth, cur = [], 0, ""
    for ch in text:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def test_ffi_is_the_generators_output():
    assert open(gen.OUT).read() == gen.render(header_functions()), "run python tools/gen_rust_ffi.py"


def test_symbol_sets_and_argument_counts_agree():
    funcs = header_functions()
    src = open(gen.OUT).read()
    fields = dict(re.findall(r'^    pub (\w+): unsafe extern "C" fn\((.*?)\)(?: -> [^,]+)?,$', src, flags=re.M))
    syms = dict(re.findall(r'^\s+(\w+): \*lib\.get::<unsafe extern "C" fn\((.*?)\)(?: -> .+?)?>\(b"(?:molar_hip_\w+)\\0"\)\?,$',
                           src, flags=re.M))
    names = re.findall(r'b"(molar_hip_\w+)\\0"', src)
    assert len(funcs) >= 60 and len(set(n for n, _, _ in funcs)) == len(funcs)
    assert sorted(names) == sorted(n for n, _, _ in funcs)
    assert set(fields) == set(syms) == {n[len("molar_hip_"):] for n, _, _ in funcs}
    for name, ret, params in funcs:
        short = name[len("molar_hip_"):]
        assert len(split_args(fields[short])) == len(params), name
        assert split_args(fields[short]) == split_args(syms[short]), name
    assert f"pub const ENTRY_POINTS: usize = {len(funcs)};" in src


def test_library_exports_exactly_the_header():
    from molar_amd import build
    lib = build.build_library()
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T molar_hip_" in ln}
    assert exported == {n for n, _, _ in header_functions()}


def test_repr_c_structs_match_the_header():
    hdr = gen.strip_comments(open(gen.HEADER).read())
    rs = open(os.path.join(CRATE, "src", "types.rs")).read()
    for cname, rname in (("molar_hip_box", "MolarHipBox"), ("molar_hip_search_desc", "MolarHipSearchDesc"),
                         ("molar_hip_search_desc_f64", "MolarHipSearchDescF64"),
                         ("molar_hip_membrane_patches", "MolarHipMembranePatches"),
                         ("molar_hip_membrane_state", "MolarHipMembraneState"),
                         ("molar_hip_membrane_desc", "MolarHipMembraneDesc"), ("molar_hip_membrane_view", "MolarHipMembraneView"),
                         ("molar_hip_membrane_out", "MolarHipMembraneOut")):
        body = re.search(r"typedef\s+struct\s*\{([^{}]*)\}\s*" + cname + r"\s*;", hdr).group(1)
        cfields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                cfields.append(re.search(r"(\w+)\s*(\[[^\]]*\])?\s*$", part.strip()).group(1))
        rbody = re.search(r"pub struct " + rname + r"\s*\{(.*?)\n\}", rs, flags=re.S).group(1)
        rfields = re.findall(r"pub (\w+):", rbody)
        assert rfields == cfields, (cname, rfields, cfields)
    assert "#[repr(C)]" in rs


def test_safe_wrappers_call_existing_entries_with_the_right_arity():
    funcs = {n[len("molar_hip_"):]: len(p) for n, _, p in header_functions()}
    src = open(os.path.join(CRATE, "src", "lib.rs")).read()
    calls = re.findall(r"\((?:self\.plugin\.fns|self\.engine\.plugin\.fns|engine\.plugin\.fns|f|plugin\.fns|self\.fns)\.(\w+)\)\(", src)
    assert len(calls) >= 15
    for m in re.finditer(r"\((?:self\.plugin\.fns|self\.engine\.plugin\.fns|engine\.plugin\.fns|f|plugin\.fns|self\.fns)\.(\w+)\)\(", src):
        name = m.group(1)
        assert name in funcs, name
        # argument list up to the matching parenthesis
        depth, k = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(src[k], 0)
            k += 1
        args = split_args(src[m.end():k - 1])
        assert len(args) == funcs[name], (name, args)
    loader = open(os.path.join(CRATE, "src", "lib.rs")).read()
    assert 'std::env::var("MOLAR_HIP_PLUGIN")' in loader and 'option_env!("MOLAR_HIP_PLUGIN")' in loader and "OnceLock" in loader
