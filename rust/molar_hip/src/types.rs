//! `#[repr(C)]` mirrors of the types of `include/molar_hip.h` (line numbers refer to that header).

use std::os::raw::c_void;

/// Opaque engine context (`molar_hip_ctx`, header :62): one HIP stream plus engine-owned device buffers.
#[repr(C)]
pub struct MolarHipCtx {
    _opaque: [u8; 0],
}

/// Opaque XTC reader (`molar_hip_xtc`).
#[repr(C)]
pub struct MolarHipXtc {
    _opaque: [u8; 0],
}

/// Opaque streamed fit (`molar_hip_fit_stream`): per-frame Kabsch fit of host-memory frames, selection packed by host threads.
#[repr(C)]
pub struct MolarHipFitStream {
    _opaque: [u8; 0],
}

/// `molar_hip_box` (header :96-101): MolAR's `PeriodicBox` (periodic_box.rs:15-23) - matrix with columns a, b, c,
/// its inverse and the triclinic correction shifts.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct MolarHipBox {
    pub m: [f32; 9],
    pub inv: [f32; 9],
    pub nshift: i32,
    pub shifts: [f32; 78],
}

/// `molar_hip_search_desc` (header :137-155): one distance-search request.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct MolarHipSearchDesc {
    pub kind: i32,
    pub cutoff: f32,
    pub xyz1: *const f32,
    pub natoms1: usize,
    pub idx1: *const u64,
    pub n1: usize,
    pub xyz2: *const f32,
    pub natoms2: usize,
    pub idx2: *const u64,
    pub n2: usize,
    pub vdw1: *const f32,
    pub vdw2: *const f32,
    pub ids_local: i32,
    pub box9: *const f32,
    pub pbc: u8,
    pub lower3: *const f32,
    pub upper3: *const f32,
}

impl Default for MolarHipSearchDesc {
    fn default() -> Self {
        // all-zero is the C side's "unset" for every field
        unsafe { std::mem::zeroed() }
    }
}

/// `molar_hip_search_desc_f64`: the same request for MolAR built with its `f64` feature (Float = f64).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct MolarHipSearchDescF64 {
    pub kind: i32,
    pub cutoff: f64,
    pub xyz1: *const f64,
    pub natoms1: usize,
    pub idx1: *const u64,
    pub n1: usize,
    pub xyz2: *const f64,
    pub natoms2: usize,
    pub idx2: *const u64,
    pub n2: usize,
    pub vdw1: *const f64,
    pub vdw2: *const f64,
    pub ids_local: i32,
    pub box9: *const f64,
    pub pbc: u8,
    pub lower3: *const f64,
    pub upper3: *const f64,
}

impl Default for MolarHipSearchDescF64 {
    fn default() -> Self {
        unsafe { std::mem::zeroed() }
    }
}

/// `molar_hip_membrane_patches`: CSR of the patch (neighbour) lists of K lipids.
#[repr(C)]
pub struct MolarHipMembranePatches {
    pub nlipids: usize,
    pub patch_offsets: *const u64,
    pub patch_ids: *const u64,
}

/// `molar_hip_membrane_state`: per-lipid arrays updated by one smoothing pass (molar_membrane/src/lib.rs:661-812).
#[repr(C)]
pub struct MolarHipMembraneState {
    pub head_markers: *mut f32,
    pub normals: *mut f32,
    pub valid: *mut u8,
    pub quad_coefs: *mut f32,
    pub mean_curv: *mut f32,
    pub gauss_curv: *mut f32,
    pub princ_curvs: *mut f32,
    pub princ_dirs: *mut f32,
    pub area: *mut f32,
    pub nvert: *mut u32,
    pub neib_ids: *mut u64,
    pub voro_vertexes: *mut f32,
    pub fitted_patch_points: *mut f32,
}

/// `molar_hip_membrane_plan`: opaque handle of the chained bilayer frame call (molar_hip_membrane_frame_*).
#[repr(C)]
pub struct MolarHipMembranePlan {
    _private: [u8; 0],
}

/// `molar_hip_membrane_desc`: what is constant over a trajectory - index lists (host memory, copied at creation) and the
/// options of Membrane::compute the chained call covers (molar_membrane/src/lib.rs:53-85, 410-454).
#[repr(C)]
pub struct MolarHipMembraneDesc {
    pub natoms: usize,
    pub nlipids: usize,
    pub lipid_idx: *const u64,
    pub lipid_offsets: *const u64,
    pub marker_idx: *const u64,
    pub marker_offsets: *const u64,
    pub masses: *const f32,
    pub ntails: usize,
    pub tail_idx: *const u64,
    pub tail_offsets: *const u64,
    pub tail_lipid: *const u32,
    pub tail_bonds: *const u8,
    pub cutoff: f32,
    pub order_type: i32,
    pub max_smooth_iter: i32,
    pub unwrap: i32,
    pub use_global_normal: i32,
    pub global_normal: [f32; 3],
}

/// `molar_hip_membrane_view`: device addresses and sizes of one frame's results.
#[repr(C)]
pub struct MolarHipMembraneView {
    pub nlipids: usize,
    pub patch_entries: usize,
    pub npairs: usize,
    pub head: *const f32,
    pub mid: *const f32,
    pub tail: *const f32,
    pub patch_offsets: *const u64,
    pub patch_ids: *const u64,
    pub initial_normals: *const f32,
    pub valid: *const u8,
    pub smoothed_head: *const f32,
    pub normals: *const f32,
    pub quad_coefs: *const f32,
    pub mean_curv: *const f32,
    pub gauss_curv: *const f32,
    pub princ_curvs: *const f32,
    pub princ_dirs: *const f32,
    pub area: *const f32,
    pub nvert: *const u32,
    pub neib_ids: *const u64,
    pub voro_vertexes: *const f32,
    pub fitted_patch_points: *const f32,
    pub order: *const f32,
    pub norder: usize,
}

/// `molar_hip_membrane_out`: host destinations of `molar_hip_membrane_frame_fetch` (null = skip).
#[repr(C)]
pub struct MolarHipMembraneOut {
    pub head: *mut f32,
    pub mid: *mut f32,
    pub tail: *mut f32,
    pub patch_offsets: *mut u64,
    pub patch_ids: *mut u64,
    pub initial_normals: *mut f32,
    pub valid: *mut u8,
    pub smoothed_head: *mut f32,
    pub normals: *mut f32,
    pub quad_coefs: *mut f32,
    pub mean_curv: *mut f32,
    pub gauss_curv: *mut f32,
    pub princ_curvs: *mut f32,
    pub princ_dirs: *mut f32,
    pub area: *mut f32,
    pub nvert: *mut u32,
    pub neib_ids: *mut u64,
    pub voro_vertexes: *mut f32,
    pub fitted_patch_points: *mut f32,
    pub order: *mut f32,
}

/// Search kinds (header :124-129) = the four driver families of distance_search.rs.
pub const SEARCH_SINGLE: i32 = 0;
pub const SEARCH_DOUBLE: i32 = 1;
pub const SEARCH_WITHIN: i32 = 2;
pub const SEARCH_DOUBLE_VDW: i32 = 3;

/// `PbcDims` bit masks (periodic_box.rs:126-128).
pub const PBC_FULL: u8 = 7;
pub const PBC_NONE: u8 = 0;

/// A raw `hipStream_t` handed to `molar_hip_set_stream`.
pub type HipStream = *mut c_void;
