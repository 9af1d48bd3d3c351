"""Loader for libmolar_hip.so (the C-ABI engine).

Search order mirrors MolAR's own plugin loader (molar_gromacs/src/lib.rs:95-116):
  1. MOLAR_HIP_PLUGIN environment variable (user override),
  2. the in-tree build next to this file (molar_amd/libmolar_hip.so).
There is deliberately NO CPU fallback: if the library is missing or no GPU is visible the
product path raises — a silent eager path would void every parity claim.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "libmolar_hip.so")

_lib = None


class MolarHipError(RuntimeError):
    """Status code + molar_hip_last_error() text.  Codes: include/molar_hip.h."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"[molar_hip status {code}] {msg}")
        self.code = code


class Box(C.Structure):
    """molar_hip_box (include/molar_hip.h) == PeriodicBox (periodic_box.rs:15-23)."""
    _fields_ = [("m", C.c_float * 9), ("inv", C.c_float * 9), ("nshift", C.c_int32), ("shifts", C.c_float * 78)]


class SearchDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("cutoff", C.c_float),
        ("xyz1", C.c_void_p), ("natoms1", C.c_size_t), ("idx1", C.c_void_p), ("n1", C.c_size_t),
        ("xyz2", C.c_void_p), ("natoms2", C.c_size_t), ("idx2", C.c_void_p), ("n2", C.c_size_t),
        ("vdw1", C.c_void_p), ("vdw2", C.c_void_p),
        ("ids_local", C.c_int32),
        ("box9", C.c_void_p), ("pbc", C.c_uint8),
        ("lower3", C.c_void_p), ("upper3", C.c_void_p),
    ]


class SearchDescF64(C.Structure):
    """molar_hip_search_desc_f64: the request of the f64 drivers (MolAR's `f64` feature)."""
    _fields_ = [
        ("kind", C.c_int32), ("cutoff", C.c_double),
        ("xyz1", C.c_void_p), ("natoms1", C.c_size_t), ("idx1", C.c_void_p), ("n1", C.c_size_t),
        ("xyz2", C.c_void_p), ("natoms2", C.c_size_t), ("idx2", C.c_void_p), ("n2", C.c_size_t),
        ("vdw1", C.c_void_p), ("vdw2", C.c_void_p),
        ("ids_local", C.c_int32),
        ("box9", C.c_void_p), ("pbc", C.c_uint8),
        ("lower3", C.c_void_p), ("upper3", C.c_void_p),
    ]


class MembraneDesc(C.Structure):
    """molar_hip_membrane_desc: what is constant over a trajectory for molar_hip_membrane_frame_*."""
    _fields_ = [
        ("natoms", C.c_size_t), ("nlipids", C.c_size_t),
        ("lipid_idx", C.c_void_p), ("lipid_offsets", C.c_void_p), ("marker_idx", C.c_void_p), ("marker_offsets", C.c_void_p),
        ("masses", C.c_void_p), ("ntails", C.c_size_t), ("tail_idx", C.c_void_p), ("tail_offsets", C.c_void_p),
        ("tail_lipid", C.c_void_p), ("tail_bonds", C.c_void_p),
        ("cutoff", C.c_float), ("order_type", C.c_int32), ("max_smooth_iter", C.c_int32), ("unwrap", C.c_int32),
        ("use_global_normal", C.c_int32), ("global_normal", C.c_float * 3),
    ]


MEMBRANE_ARRAYS = ("head", "mid", "tail", "patch_offsets", "patch_ids", "initial_normals", "valid", "smoothed_head", "normals",
                   "quad_coefs", "mean_curv", "gauss_curv", "princ_curvs", "princ_dirs", "area", "nvert", "neib_ids",
                   "voro_vertexes", "fitted_patch_points", "order")


class MembraneView(C.Structure):
    """molar_hip_membrane_view: device addresses of one frame's results."""
    _fields_ = ([("nlipids", C.c_size_t), ("patch_entries", C.c_size_t), ("npairs", C.c_size_t)]
                + [(k, C.c_void_p) for k in MEMBRANE_ARRAYS] + [("norder", C.c_size_t)])


class MembraneOut(C.Structure):
    """molar_hip_membrane_out: host destinations of molar_hip_membrane_frame_fetch."""
    _fields_ = [(k, C.c_void_p) for k in MEMBRANE_ARRAYS]


# every symbol include/molar_hip.h declares: name -> (restype, argtypes)
_P, _SZ, _F, _I, _U8 = C.c_void_p, C.c_size_t, C.c_float, C.c_int, C.c_uint8
SYMBOLS = {
    "molar_hip_create": (C.c_void_p, [_I]),
    "molar_hip_destroy": (None, [_P]),
    "molar_hip_last_error": (C.c_char_p, []),
    "molar_hip_version": (C.c_char_p, []),
    "molar_hip_device_count": (_I, []),
    "molar_hip_set_stream": (_I, [_P, _P]),
    "molar_hip_synchronize": (_I, [_P]),
    "molar_hip_profile_enable": (_I, [_P, _I]),
    "molar_hip_profile_read": (_I, [_P, _P, _P]),
    "molar_hip_box_from_matrix": (_I, [_P, _P]),
    "molar_hip_box_from_vectors_angles": (_I, [_F, _F, _F, _F, _F, _F, _P]),
    "molar_hip_box_shortest_vector": (None, [_P, _P, _U8, _P]),
    "molar_hip_copy_bandwidth": (_I, [_P, _SZ, _I, _P]),
    "molar_hip_write_bandwidth": (_I, [_P, _SZ, _I, _P]),
    "molar_hip_box_lab_extents": (None, [_P, _P]),
    "molar_hip_box_extents": (None, [_P, _P]),
    "molar_hip_box_to_box_coords": (None, [_P, _P, _P]),
    "molar_hip_box_to_lab_coords": (None, [_P, _P, _P]),
    "molar_hip_box_is_inside": (_I, [_P, _P]),
    "molar_hip_box_wrap_point": (None, [_P, _P, _P]),
    "molar_hip_search_count": (_I, [_P, _P, _P]),
    "molar_hip_search_fill": (_I, [_P, _P, _P]),
    "molar_hip_search_fill_usize": (_I, [_P, _P, _P, _P]),
    "molar_hip_search_fill_ids": (_I, [_P, _P]),
    "molar_hip_within_count": (_I, [_P, _P, _P]),
    "molar_hip_within_fill": (_I, [_P, _P]),
    "molar_hip_within_hold": (_I, [_P, _I]),
    "molar_hip_search_grid_dims": (_I, [_P, _P]),
    "molar_hip_search_cell_kernels": (_I, [_P, _P, _P]),
    "molar_hip_search_count_f64": (_I, [_P, _P, _P]),
    "molar_hip_search_fill_f64": (_I, [_P, _P, _P, _P]),
    "molar_hip_search_fill_ids_f64": (_I, [_P, _P]),
    "molar_hip_search_grid_dims_f64": (_I, [_P, _P]),
    "molar_hip_search_resident": (_I, [_P, _P, _P, _P, _P]),
    "molar_hip_search_fill_device": (_I, [_P, _P, _P]),
    "molar_hip_search_resident_planes": (_I, [_P, _I]),
    "molar_hip_search_resident_begin": (_I, [_P, _P, _P]),
    "molar_hip_search_resident_end": (_I, [_P, C.c_int32, _P, _P, _P]),
    "molar_hip_search_histogram": (_I, [_P, _P, _F, _F, _SZ, _P, _P]),
    "molar_hip_search_histogram_frames": (_I, [_P, _P, _SZ, _SZ, _SZ, _P, _F, _F, _SZ, _P]),
    "molar_hip_histogram_edges": (_I, [_F, _F, _SZ, _P]),
    "molar_hip_min_max": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P]),
    "molar_hip_center_of_geometry": (_I, [_P, _P, _SZ, _P, _SZ, _P]),
    "molar_hip_center_of_mass": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P]),
    "molar_hip_center_of_geometry_pbc": (_I, [_P, _P, _SZ, _P, _SZ, _P, _U8, _P]),
    "molar_hip_center_of_mass_pbc": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P, _U8, _P]),
    "molar_hip_gyration": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P, _P]),
    "molar_hip_inertia": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P, _P, _P, _P]),
    "molar_hip_rmsd": (_I, [_P, _P, _SZ, _P, _SZ, _P, _SZ, _P, _SZ, _P]),
    "molar_hip_rmsd_mw": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P, _SZ, _P, _SZ, _P]),
    "molar_hip_fit_transform": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P, _SZ, _P, _SZ, _P, _I, _P, _P]),
    "molar_hip_xtc_histogram": (_I, [_P, _P, _SZ, _SZ, _P, _SZ, _F, _U8, _F, _F, _SZ, _P, _I]),
    "molar_hip_xtc_histogram_double": (_I, [_P, _P, _SZ, _SZ, _P, _SZ, _P, _SZ, _F, _U8, _F, _F, _SZ, _P, _I]),
    "molar_hip_fit_stream_create": (_I, [_P, _SZ, _P, _SZ, _P, _P, _SZ, _P, _I, _P]),
    "molar_hip_fit_stream_begin": (_I, [_P, _P, _I, _P]),
    "molar_hip_fit_stream_end": (_I, [_P, C.c_int32, _P, _P, _P, _P, _P]),
    "molar_hip_fit_stream_destroy": (None, [_P]),
    "molar_hip_center_of_geometry_f64": (_I, [_P, _P, _SZ, _P, _SZ, _P]),
    "molar_hip_center_of_mass_f64": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P]),
    "molar_hip_gyration_f64": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P]),
    "molar_hip_rmsd_f64": (_I, [_P, _P, _SZ, _P, _SZ, _P, _SZ, _P, _SZ, _P]),
    "molar_hip_rmsd_mw_f64": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P, _SZ, _P, _SZ, _P]),
    "molar_hip_fit_transform_f64": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P, _SZ, _P, _SZ, _P, _I, _P, _P]),
    "molar_hip_apply_transform_f64": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P]),
    "molar_hip_fit_rmsd_batch_f64": (_I, [_P, _P, _SZ, _SZ, _P, _SZ, _P, _P, _SZ, _P, _I, _P, _P, _P, _P, _P]),
    "molar_hip_center_of_geometry_pbc_f64": (_I, [_P, _P, _SZ, _P, _SZ, _P, _U8, _P]),
    "molar_hip_center_of_mass_pbc_f64": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P, _U8, _P]),
    "molar_hip_gyration_pbc_f64": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P, _P]),
    "molar_hip_unwrap_simple_f64": (_I, [_P, _P, _SZ, _P, _SZ, _P, _U8]),
    "molar_hip_inertia_pbc_f64": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P, _P, _P, _P]),
    "molar_hip_rotate_f64": (_I, [_P, _P, _SZ, _P, _SZ, _P, C.c_double]),
    "molar_hip_principal_transform_f64": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P, _P, _P]),
    "molar_hip_lipid_tail_order_f64": (_I, [_P, _P, _SZ, _P, _P, _SZ, _I, _P, _P, _P, _P]),
    "molar_hip_min_max_f64": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P]),
    "molar_hip_inertia_f64": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P, _P, _P]),
    "molar_hip_translate_f64": (_I, [_P, _P, _SZ, _P, _SZ, _P]),
    "molar_hip_center_batch": (_I, [_P, _P, _SZ, _P, _P, _SZ, _P, _P]),
    "molar_hip_unwrap_simple_batch": (_I, [_P, _P, _SZ, _P, _P, _SZ, _P, _U8]),
    "molar_hip_membrane_initial_normals": (_I, [_SZ, _P, _P, _P, _P, _P, _P]),
    "molar_hip_membrane_smooth": (_I, [_P, _P, _P, _P]),
    "molar_hip_membrane_patches_from_pairs": (_I, [_P, _SZ, _SZ, _P, _P]),
    "molar_hip_membrane_nth_shell_patches": (_I, [_SZ, _P, _P, _P, _P, _P, _SZ, _P, _P, _SZ, _P]),
    "molar_hip_membrane_smooth_curvature": (_I, [_SZ, _P, _P, _P, _P, _SZ, _P, _P]),
    "molar_hip_membrane_plan_create": (_I, [_P, _P, _P]),
    "molar_hip_membrane_plan_destroy": (None, [_P]),
    "molar_hip_membrane_plan_set_valid": (_I, [_P, _P]),
    "molar_hip_membrane_frame_begin": (_I, [_P, _P, _P, _P]),
    "molar_hip_membrane_frame_end": (_I, [_P, _I, _P]),
    "molar_hip_membrane_frame_fetch": (_I, [_P, _I, _P]),
    "molar_hip_membrane_frame_end_fetch": (_I, [_P, _I, _P, _P]),
    "molar_hip_xtc_open": (_P, [C.c_char_p]),
    "molar_hip_xtc_open_memory": (_P, [_P, _SZ]),
    "molar_hip_xtc_close": (None, [_P]),
    "molar_hip_xtc_nframes": (_SZ, [_P]),
    "molar_hip_xtc_natoms": (_SZ, [_P]),
    "molar_hip_xtc_frame_info": (_I, [_P, _SZ, _P, _P, _P, _P, _P]),
    "molar_hip_xtc_seek_time": (_I, [_P, C.c_float, _P]),
    "molar_hip_xtc_read": (_I, [_P, _P, _SZ, _SZ, _P, _I]),
    "molar_hip_xtc_read_device": (_I, [_P, _P, _SZ, _SZ, _P]),
    "molar_hip_xtc_encode_frame": (_I, [_P, _SZ, _P, C.c_int32, C.c_float, C.c_float, _P, _SZ, _P]),
    "molar_hip_lipid_tail_order": (_I, [_P, _P, _SZ, _P, _P, _SZ, _I, _P, _P, _P, _P]),
    "molar_hip_apply_transform": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P]),
    "molar_hip_unwrap_simple": (_I, [_P, _P, _SZ, _P, _SZ, _P, _U8]),
    "molar_hip_search_connectivity": (_I, [_P, _P, _P, _P]),
    "molar_hip_search_connectivity_fill": (_I, [_P, _P, _P]),
    "molar_hip_unwrap_connectivity": (_I, [_P, _P, _SZ, _P, _SZ, _P, _F, _U8, _P, _P, _P]),
    "molar_hip_fit_rmsd_batch": (_I, [_P, _P, _SZ, _SZ, _P, _SZ, _P, _P, _SZ, _P, _I, _P, _P, _P, _P, _P]),
    "molar_hip_gyration_batch": (_I, [_P, _P, _SZ, _P, _P, _SZ, _P, _P, _P]),
    "molar_hip_rmsd_batch": (_I, [_P, _P, _SZ, _P, _P, _SZ, _P, _P, _SZ, _P, _P]),
    "molar_hip_fit_batch": (_I, [_P, _P, _SZ, _P, _P, _P, _SZ, _P, _P, _P, _SZ, _I, _P, _P, _P, _P, _P]),
    "molar_hip_translate": (_I, [_P, _P, _SZ, _P, _SZ, _P]),
    "molar_hip_rotate": (_I, [_P, _P, _SZ, _P, _SZ, _P, _F]),
    "molar_hip_principal_transform": (_I, [_P, _P, _SZ, _P, _SZ, _P, _P, _P, _P]),
}


def lib_path() -> str:
    return os.environ.get("MOLAR_HIP_PLUGIN", DEFAULT_LIB)


def _preload_torch_hip_runtime():
    """torch wheels bundle their own libamdhip64.so.7 (same SONAME as /opt/rocm's).  Whichever
    copy is dlopen'ed first serves the whole process; torch does not find its GPUs behind the
    system copy, so when torch is installed load its runtime first.  (A non-Python host simply
    uses the system runtime the library is linked against.)"""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen the engine and bind every symbol of the header; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    _preload_torch_hip_runtime()
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"libmolar_hip.so not found at {path}: build it with `python -m molar_amd.build` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().molar_hip_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != 0:
        raise MolarHipError(rc, last_error())
