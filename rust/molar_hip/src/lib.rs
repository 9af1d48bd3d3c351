//! Host shim between MolAR and the MI355X engine (`libmolar_hip.so`).
//!
//! **Source only**: this crate has not been compiled in the repository's build image (no `cargo` / `rustc` there).
//! `src/ffi.rs` is generated from `include/molar_hip.h` by `tools/gen_rust_ffi.py`, and
//! `tests/test_rust_shim_cpu.py` asserts that the symbol set, the argument counts and the safe wrappers below stay in
//! step with the C header.
//!
//! What MolAR hands over is exactly what its provider traits already hold:
//!
//! | MolAR                                                   | this crate                         |
//! |---------------------------------------------------------|------------------------------------|
//! | `PosProvider::coords_ptr()` (providers.rs:96-106)       | `&[[f32; 3]]`, the WHOLE frame     |
//! | `IndexSliceProvider::get_index_slice()` (:45-76)        | `Option<&[usize]>` (None = all)    |
//! | `AtomStorage::masses()` (atom_storage.rs:272)           | `&[f32]`, full-length column       |
//! | `PeriodicBox` matrix, columns a,b,c (periodic_box.rs)   | `&[f32; 9]` column-major           |
//! | `PbcDims` (periodic_box.rs:70-128)                      | `u8` mask                          |
//!
//! Loading follows MolAR's own plugin crate (molar_gromacs/src/lib.rs:95-176): `MOLAR_HIP_PLUGIN` at run time, then
//! the path baked in at compile time, then the system search path; symbols are resolved once into a table of function
//! pointers; the loaded plugin is cached process-wide.  When no library (or no GPU) is found the caller keeps MolAR's
//! CPU path - `Engine::try_default()` returns `None`, nothing else changes.

pub mod ffi;
pub mod types;

use std::ffi::CStr;
use std::sync::{Arc, OnceLock};

use ffi::MolarHipFns;
pub use types::*;

/// MolAR's `Float` (molar/src/aliases.rs:10-13).
#[cfg(not(feature = "f64"))]
pub type Float = f32;
#[cfg(feature = "f64")]
pub type Float = f64;

/// Errors of the engine, mapped from the C status codes (header :38-55) onto MolAR's error enums.
#[derive(thiserror::Error, Debug)]
pub enum EngineError {
    #[error("engine library not available: {0}")]
    Load(#[from] libloading::Error),
    /// `MeasureError::Sizes` (measure.rs:732-762)
    #[error("incompatible sizes: {0}")]
    Sizes(String),
    /// `MeasureError::ZeroMass`
    #[error("zero mass")]
    ZeroMass,
    /// `MeasureError::Svd`
    #[error("SVD failed")]
    Svd,
    /// `MeasureError::NoPbc`
    #[error("pbc operation without periodic box")]
    NoPbc,
    /// `PeriodicBoxError` (periodic_box.rs:131-144)
    #[error("periodic box: {0}")]
    PeriodicBox(String),
    /// `LipidOrderError` (measure.rs:281-291)
    #[error("lipid order: {0}")]
    LipidOrder(String),
    #[error("engine: {0}")]
    Other(String),
}

/// The loaded library plus its resolved entry points.  The `Library` must outlive every call through the table, which
/// holding both in one struct guarantees.
pub struct Plugin {
    _lib: libloading::Library,
    pub fns: MolarHipFns,
}

// The table is plain function pointers; contexts are what must not be shared between threads (see `Engine`).
unsafe impl Send for Plugin {}
unsafe impl Sync for Plugin {}

impl Plugin {
    fn open_library() -> Result<libloading::Library, libloading::Error> {
        // 1. user override at run time
        if let Ok(path) = std::env::var("MOLAR_HIP_PLUGIN") {
            return unsafe { libloading::Library::new(path) };
        }
        // 2. location recorded when the crate was built
        if let Some(path) = option_env!("MOLAR_HIP_PLUGIN") {
            if let Ok(lib) = unsafe { libloading::Library::new(path) } {
                return Ok(lib);
            }
        }
        // 3. system search path
        unsafe { libloading::Library::new(libloading::library_filename("molar_hip")) }
    }

    pub fn load() -> Result<Self, libloading::Error> {
        let lib = Self::open_library()?;
        let fns = unsafe { MolarHipFns::resolve(&lib)? };
        Ok(Plugin { _lib: lib, fns })
    }

    /// Process-wide instance; loaded at most once, a failed attempt is retried by the next call.
    pub fn get_cached() -> Result<Arc<Self>, libloading::Error> {
        static CACHE: OnceLock<Arc<Plugin>> = OnceLock::new();
        if let Some(p) = CACHE.get() {
            return Ok(Arc::clone(p));
        }
        let fresh = Arc::new(Self::load()?);
        let _ = CACHE.set(Arc::clone(&fresh));
        Ok(CACHE.get().map(Arc::clone).unwrap_or(fresh))
    }

    fn last_error(&self) -> String {
        unsafe {
            let p = (self.fns.last_error)();
            if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
        }
    }

    fn check(&self, status: i32) -> Result<(), EngineError> {
        match status {
            0 => Ok(()),
            1 => Err(EngineError::Sizes(self.last_error())),
            2 => Err(EngineError::ZeroMass),
            3 => Err(EngineError::Svd),
            4 => Err(EngineError::NoPbc),
            5 | 6 | 10 => Err(EngineError::PeriodicBox(self.last_error())),
            7..=9 => Err(EngineError::LipidOrder(self.last_error())),
            _ => Err(EngineError::Other(self.last_error())),
        }
    }
}

/// One engine context = one HIP stream with its device buffers.  Not `Sync`: MolAR calls `Measure` from rayon workers
/// on disjoint selections (system.rs:193-213) - give each worker its own `Engine`, or use the `*_batch` calls.
pub struct Engine {
    plugin: Arc<Plugin>,
    ctx: *mut MolarHipCtx,
}

unsafe impl Send for Engine {}

impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { (self.plugin.fns.destroy)(self.ctx) }
    }
}

fn idx_ptr(idx: Option<&[usize]>) -> (*const u64, usize) {
    // usize == u64 on every target MolAR supports (aliases.rs); the engine reads u64
    match idx {
        Some(s) => (s.as_ptr() as *const u64, s.len()),
        None => (std::ptr::null(), 0),
    }
}

// The C ABI trusts its pointer/size pairs.  Safe callers cannot be allowed to hand it a mass column shorter than the
// frame, an index past the end of the coordinates or a CSR split that leaves its index slice: the wrappers below check
// exactly what the engine will dereference and return `EngineError::Sizes` instead of reading out of bounds.
fn check_index(index: Option<&[usize]>, natoms: usize, what: &str) -> Result<(), EngineError> {
    if let Some(ix) = index {
        if let Some(&bad) = ix.iter().find(|&&a| a >= natoms) {
            return Err(EngineError::Sizes(format!("{what}: index {bad} is out of range for {natoms} atoms")));
        }
    }
    Ok(())
}

fn check_column(len: usize, natoms: usize, what: &str) -> Result<(), EngineError> {
    // per-atom columns (masses) are gathered through the index: they must cover the whole frame
    if len < natoms {
        return Err(EngineError::Sizes(format!("{what}: column of {len} entries for {natoms} atoms")));
    }
    Ok(())
}

fn check_offsets(offsets: &[usize], index_len: usize, what: &str) -> Result<(), EngineError> {
    if offsets.windows(2).any(|w| w[0] > w[1]) || offsets.last().map_or(false, |&e| e > index_len) {
        return Err(EngineError::Sizes(format!("{what}: offsets must be non-decreasing and end within the {index_len} indices")));
    }
    Ok(())
}

fn box_ptr(b: Option<&[f32; 9]>) -> *const f32 {
    b.map_or(std::ptr::null(), |m| m.as_ptr())
}

impl Engine {
    pub fn new(device: i32) -> Result<Self, EngineError> {
        let plugin = Plugin::get_cached()?;
        let ctx = unsafe { (plugin.fns.create)(device) };
        if ctx.is_null() {
            return Err(EngineError::Other(plugin.last_error()));
        }
        Ok(Engine { plugin, ctx })
    }

    /// `Some(engine)` on a machine with the library and a GPU, `None` otherwise: the caller stays on MolAR's CPU path.
    pub fn try_default() -> Option<Self> {
        Self::new(0).ok()
    }

    pub fn synchronize(&self) -> Result<(), EngineError> {
        self.plugin.check(unsafe { (self.plugin.fns.synchronize)(self.ctx) })
    }

    // ---------------------------------------------------------------- distance search (distance_search.rs:519-954)

    fn search(&self, d: &MolarHipSearchDesc) -> Result<Vec<(usize, usize, Float)>, EngineError> {
        let f = &self.plugin.fns;
        let mut n = 0u64;
        self.plugin.check(unsafe { (f.search_count)(self.ctx, d, &mut n) })?;
        let n = n as usize;
        let (mut i, mut j, mut dist) = (vec![0u64; n], vec![0u64; n], vec![0f32; n]);
        self.plugin.check(unsafe { (f.search_fill_usize)(self.ctx, i.as_mut_ptr(), j.as_mut_ptr(), dist.as_mut_ptr()) })?;
        Ok((0..n).map(|k| (i[k] as usize, j[k] as usize, dist[k] as Float)).collect())
    }

    /// `distance_search_single_pbc` (distance_search.rs:928-954); `box9 = None` is `distance_search_single` (:892-926).
    /// Result order is the reference's: plan order, then i-major / j-minor.
    pub fn distance_search_single(
        &self, cutoff: f32, coords: &[[f32; 3]], index: Option<&[usize]>, box9: Option<&[f32; 9]>, pbc: u8,
    ) -> Result<Vec<(usize, usize, Float)>, EngineError> {
        check_index(index, coords.len(), "distance_search_single")?;
        let (ip, n) = idx_ptr(index);
        let d = MolarHipSearchDesc {
            kind: SEARCH_SINGLE, cutoff, xyz1: coords.as_ptr() as *const f32, natoms1: coords.len(), idx1: ip, n1: n,
            box9: box_ptr(box9), pbc, ..Default::default()
        };
        self.search(&d)
    }

    /// `distance_search_double(_pbc)` (:659-754): pairs (atom of set 1, atom of set 2).
    pub fn distance_search_double(
        &self, cutoff: f32, coords1: &[[f32; 3]], index1: Option<&[usize]>, coords2: &[[f32; 3]], index2: Option<&[usize]>,
        box9: Option<&[f32; 9]>, pbc: u8,
    ) -> Result<Vec<(usize, usize, Float)>, EngineError> {
        check_index(index1, coords1.len(), "distance_search_double (set 1)")?;
        check_index(index2, coords2.len(), "distance_search_double (set 2)")?;
        let (i1, n1) = idx_ptr(index1);
        let (i2, n2) = idx_ptr(index2);
        let d = MolarHipSearchDesc {
            kind: SEARCH_DOUBLE, cutoff, xyz1: coords1.as_ptr() as *const f32, natoms1: coords1.len(), idx1: i1, n1,
            xyz2: coords2.as_ptr() as *const f32, natoms2: coords2.len(), idx2: i2, n2, box9: box_ptr(box9), pbc,
            ..Default::default()
        };
        self.search(&d)
    }

    /// `distance_search_within_pbc` (:519-598): ids of set-1 atoms within `cutoff` of set 2, duplicates as the reference
    /// emits them (the caller sorts and de-duplicates, selection_expr.rs:112).
    pub fn distance_search_within_pbc(
        &self, cutoff: f32, coords1: &[[f32; 3]], index1: Option<&[usize]>, coords2: &[[f32; 3]], index2: Option<&[usize]>,
        box9: &[f32; 9], pbc: u8,
    ) -> Result<Vec<usize>, EngineError> {
        check_index(index1, coords1.len(), "distance_search_within_pbc (set 1)")?;
        check_index(index2, coords2.len(), "distance_search_within_pbc (set 2)")?;
        let (i1, n1) = idx_ptr(index1);
        let (i2, n2) = idx_ptr(index2);
        let d = MolarHipSearchDesc {
            kind: SEARCH_WITHIN, cutoff, xyz1: coords1.as_ptr() as *const f32, natoms1: coords1.len(), idx1: i1, n1,
            xyz2: coords2.as_ptr() as *const f32, natoms2: coords2.len(), idx2: i2, n2, box9: box9.as_ptr(), pbc,
            ..Default::default()
        };
        let f = &self.plugin.fns;
        let mut n = 0u64;
        self.plugin.check(unsafe { (f.search_count)(self.ctx, &d, &mut n) })?;
        let mut ids = vec![0u64; n as usize];
        self.plugin.check(unsafe { (f.search_fill_ids)(self.ctx, ids.as_mut_ptr()) })?;
        Ok(ids.into_iter().map(|v| v as usize).collect())
    }

    /// `within <cutoff> pbc of <inner>` as the selection keeps it (`LogicalNode::Within`, selection/ast.rs:589-631): what
    /// `SortedSet::from_unsorted` (selection_expr.rs:112) makes of the stream above - sorted, de-duplicated - computed
    /// without the stream: an atom stops looking at its first hit in any plan entry (molar_hip_within_count / _fill).
    pub fn within_set_pbc(
        &self, cutoff: f32, coords1: &[[f32; 3]], index1: Option<&[usize]>, coords2: &[[f32; 3]], index2: Option<&[usize]>,
        box9: &[f32; 9], pbc: u8,
    ) -> Result<Vec<usize>, EngineError> {
        check_index(index1, coords1.len(), "within_set_pbc (set 1)")?;
        check_index(index2, coords2.len(), "within_set_pbc (set 2)")?;
        let (i1, n1) = idx_ptr(index1);
        let (i2, n2) = idx_ptr(index2);
        let d = MolarHipSearchDesc {
            kind: SEARCH_WITHIN, cutoff, xyz1: coords1.as_ptr() as *const f32, natoms1: coords1.len(), idx1: i1, n1,
            xyz2: coords2.as_ptr() as *const f32, natoms2: coords2.len(), idx2: i2, n2, box9: box9.as_ptr(), pbc,
            ..Default::default()
        };
        let f = &self.plugin.fns;
        let mut n = 0u64;
        self.plugin.check(unsafe { (f.within_count)(self.ctx, &d, &mut n) })?;
        let mut ids = vec![0u64; n as usize];
        if n > 0 {
            self.plugin.check(unsafe { (f.within_fill)(self.ctx, ids.as_mut_ptr()) })?;
        }
        Ok(ids.into_iter().map(|v| v as usize).collect())
    }

    /// While a caller holds `&State` the first set of its `within` requests cannot change: with the hold on, consecutive
    /// `within_set_pbc` calls that name the same first set (same slices, same box) and come to the same grid reuse its staged
    /// coordinates and its grid (molar_hip_within_hold; the reference's within_size_bench.rs asks 1600 times against one frame).
    pub fn within_hold(&self, on: bool) -> Result<(), EngineError> {
        self.plugin.check(unsafe { (self.plugin.fns.within_hold)(self.ctx, if on { 1 } else { 0 }) })
    }

    /// `SearchConnectivity::from_iter(distance_search_single_pbc(..))` (connectivity.rs:19-35, modify.rs:77-78) built on the
    /// device: adjacency lists in the reference's push order as CSR over local ids (`offsets[len + 1]`, `neigh[2 * pairs]`).
    pub fn search_connectivity_pbc(
        &self, cutoff: f32, coords: &[[f32; 3]], index: Option<&[usize]>, box9: &[f32; 9], pbc: u8,
    ) -> Result<(Vec<usize>, Vec<usize>), EngineError> {
        check_index(index, coords.len(), "search_connectivity_pbc")?;
        let (ip, n) = idx_ptr(index);
        let d = MolarHipSearchDesc {
            kind: SEARCH_SINGLE, cutoff, xyz1: coords.as_ptr() as *const f32, natoms1: coords.len(), idx1: ip, n1: n,
            ids_local: 1, box9: box9.as_ptr(), pbc, ..Default::default()
        };
        let f = &self.plugin.fns;
        let (mut rows, mut entries) = (0u64, 0u64);
        self.plugin.check(unsafe { (f.search_connectivity)(self.ctx, &d, &mut rows, &mut entries) })?;
        let mut off = vec![0u64; rows as usize + 1];
        let mut nb = vec![0u64; entries as usize];
        self.plugin.check(unsafe { (f.search_connectivity_fill)(self.ctx, off.as_mut_ptr(), if entries > 0 { nb.as_mut_ptr() } else { std::ptr::null_mut() }) })?;
        Ok((off.into_iter().map(|v| v as usize).collect(), nb.into_iter().map(|v| v as usize).collect()))
    }

    // ---------------------------------------------------------------- Measure (measure.rs:22-649)

    /// `Measure::center_of_mass` (:60-75)
    pub fn center_of_mass(&self, coords: &[[f32; 3]], index: Option<&[usize]>, masses: &[f32]) -> Result<[f32; 3], EngineError> {
        check_index(index, coords.len(), "center_of_mass")?;
        check_column(masses.len(), coords.len(), "center_of_mass: masses")?;
        let (ip, n) = idx_ptr(index);
        let mut out = [0f32; 3];
        self.plugin.check(unsafe {
            (self.plugin.fns.center_of_mass)(self.ctx, coords.as_ptr() as *const f32, coords.len(), ip, n, masses.as_ptr(), out.as_mut_ptr())
        })?;
        Ok(out)
    }

    /// `Measure::gyration` (:78-87) / `gyration_pbc` (:222-232) with a box
    pub fn gyration(&self, coords: &[[f32; 3]], index: Option<&[usize]>, masses: &[f32], box9: Option<&[f32; 9]>) -> Result<f32, EngineError> {
        check_index(index, coords.len(), "gyration")?;
        check_column(masses.len(), coords.len(), "gyration: masses")?;
        let (ip, n) = idx_ptr(index);
        let mut out = 0f32;
        self.plugin.check(unsafe {
            (self.plugin.fns.gyration)(self.ctx, coords.as_ptr() as *const f32, coords.len(), ip, n, masses.as_ptr(), box_ptr(box9), &mut out)
        })?;
        Ok(out)
    }

    /// `rmsd` (:485-504)
    pub fn rmsd(&self, c1: &[[f32; 3]], i1: Option<&[usize]>, c2: &[[f32; 3]], i2: Option<&[usize]>) -> Result<f32, EngineError> {
        check_index(i1, c1.len(), "rmsd (selection 1)")?;
        check_index(i2, c2.len(), "rmsd (selection 2)")?;
        let (p1, n1) = idx_ptr(i1);
        let (p2, n2) = idx_ptr(i2);
        let mut out = 0f32;
        self.plugin.check(unsafe {
            (self.plugin.fns.rmsd)(self.ctx, c1.as_ptr() as *const f32, c1.len(), p1, n1, c2.as_ptr() as *const f32, c2.len(), p2, n2, &mut out)
        })?;
        Ok(out)
    }

    /// `fit_transform` (:507-522): (R column-major, t) with p -> R p + t moving selection 1 onto selection 2.
    pub fn fit_transform(
        &self, c1: &[[f32; 3]], i1: Option<&[usize]>, m1: &[f32], c2: &[[f32; 3]], i2: Option<&[usize]>, m2: &[f32],
    ) -> Result<([f32; 9], [f32; 3]), EngineError> {
        check_index(i1, c1.len(), "fit_transform (selection 1)")?;
        check_index(i2, c2.len(), "fit_transform (selection 2)")?;
        check_column(m1.len(), c1.len(), "fit_transform: masses 1")?;
        check_column(m2.len(), c2.len(), "fit_transform: masses 2")?;
        let (p1, n1) = idx_ptr(i1);
        let (p2, n2) = idx_ptr(i2);
        let (mut r, mut t) = ([0f32; 9], [0f32; 3]);
        self.plugin.check(unsafe {
            (self.plugin.fns.fit_transform)(self.ctx, c1.as_ptr() as *const f32, c1.len(), p1, n1, m1.as_ptr(),
                                            c2.as_ptr() as *const f32, c2.len(), p2, n2, m2.as_ptr(), 0, r.as_mut_ptr(), t.as_mut_ptr())
        })?;
        Ok((r, t))
    }

    /// `Modify::apply_transform` (modify.rs:32-36), in place
    pub fn apply_transform(&self, coords: &mut [[f32; 3]], index: Option<&[usize]>, r: &[f32; 9], t: &[f32; 3]) -> Result<(), EngineError> {
        check_index(index, coords.len(), "apply_transform")?;
        let (ip, n) = idx_ptr(index);
        self.plugin.check(unsafe {
            (self.plugin.fns.apply_transform)(self.ctx, coords.as_mut_ptr() as *mut f32, coords.len(), ip, n, r.as_ptr(), t.as_ptr())
        })
    }

    // ---- MolAR built with `f64` (Float = f64): the Measure / Modify methods on double-precision data
    // (header: "MolAR built with its `f64` feature"); the periodic ones are bound in ffi.rs.  The search takes f32 only.

    /// `Measure::center_of_mass` (:60-75), f64
    pub fn center_of_mass_f64(&self, coords: &[[f64; 3]], index: Option<&[usize]>, masses: &[f64]) -> Result<[f64; 3], EngineError> {
        check_index(index, coords.len(), "center_of_mass_f64")?;
        check_column(masses.len(), coords.len(), "center_of_mass_f64: masses")?;
        let (ip, n) = idx_ptr(index);
        let mut out = [0f64; 3];
        self.plugin.check(unsafe {
            (self.plugin.fns.center_of_mass_f64)(self.ctx, coords.as_ptr() as *const f64, coords.len(), ip, n, masses.as_ptr(), out.as_mut_ptr())
        })?;
        Ok(out)
    }

    /// `Measure::gyration` (:78-87), f64
    pub fn gyration_f64(&self, coords: &[[f64; 3]], index: Option<&[usize]>, masses: &[f64]) -> Result<f64, EngineError> {
        check_index(index, coords.len(), "gyration_f64")?;
        check_column(masses.len(), coords.len(), "gyration_f64: masses")?;
        let (ip, n) = idx_ptr(index);
        let mut out = 0f64;
        self.plugin.check(unsafe {
            (self.plugin.fns.gyration_f64)(self.ctx, coords.as_ptr() as *const f64, coords.len(), ip, n, masses.as_ptr(), &mut out)
        })?;
        Ok(out)
    }

    /// `rmsd` (:485-504), f64
    pub fn rmsd_f64(&self, c1: &[[f64; 3]], i1: Option<&[usize]>, c2: &[[f64; 3]], i2: Option<&[usize]>) -> Result<f64, EngineError> {
        check_index(i1, c1.len(), "rmsd_f64 (selection 1)")?;
        check_index(i2, c2.len(), "rmsd_f64 (selection 2)")?;
        let (p1, n1) = idx_ptr(i1);
        let (p2, n2) = idx_ptr(i2);
        let mut out = 0f64;
        self.plugin.check(unsafe {
            (self.plugin.fns.rmsd_f64)(self.ctx, c1.as_ptr() as *const f64, c1.len(), p1, n1, c2.as_ptr() as *const f64, c2.len(), p2, n2, &mut out)
        })?;
        Ok(out)
    }

    /// `fit_transform` (:507-522), f64: (R column-major, t)
    pub fn fit_transform_f64(
        &self, c1: &[[f64; 3]], i1: Option<&[usize]>, m1: &[f64], c2: &[[f64; 3]], i2: Option<&[usize]>, m2: &[f64],
    ) -> Result<([f64; 9], [f64; 3]), EngineError> {
        check_index(i1, c1.len(), "fit_transform_f64 (selection 1)")?;
        check_index(i2, c2.len(), "fit_transform_f64 (selection 2)")?;
        check_column(m1.len(), c1.len(), "fit_transform_f64: masses 1")?;
        check_column(m2.len(), c2.len(), "fit_transform_f64: masses 2")?;
        let (p1, n1) = idx_ptr(i1);
        let (p2, n2) = idx_ptr(i2);
        let (mut r, mut t) = ([0f64; 9], [0f64; 3]);
        self.plugin.check(unsafe {
            (self.plugin.fns.fit_transform_f64)(self.ctx, c1.as_ptr() as *const f64, c1.len(), p1, n1, m1.as_ptr(),
                                                c2.as_ptr() as *const f64, c2.len(), p2, n2, m2.as_ptr(), 0, r.as_mut_ptr(), t.as_mut_ptr())
        })?;
        Ok((r, t))
    }

    /// `Modify::apply_transform` (modify.rs:32-36), f64, in place
    pub fn apply_transform_f64(&self, coords: &mut [[f64; 3]], index: Option<&[usize]>, r: &[f64; 9], t: &[f64; 3]) -> Result<(), EngineError> {
        check_index(index, coords.len(), "apply_transform_f64")?;
        let (ip, n) = idx_ptr(index);
        self.plugin.check(unsafe {
            (self.plugin.fns.apply_transform_f64)(self.ctx, coords.as_mut_ptr() as *mut f64, coords.len(), ip, n, r.as_ptr(), t.as_ptr())
        })
    }

    /// `Modify::unwrap_simple_dim` (modify.rs:40-54), in place
    pub fn unwrap_simple_dim(&self, coords: &mut [[f32; 3]], index: Option<&[usize]>, box9: &[f32; 9], dims: u8) -> Result<(), EngineError> {
        check_index(index, coords.len(), "unwrap_simple_dim")?;
        let (ip, n) = idx_ptr(index);
        self.plugin.check(unsafe {
            (self.plugin.fns.unwrap_simple)(self.ctx, coords.as_mut_ptr() as *mut f32, coords.len(), ip, n, box9.as_ptr(), dims)
        })
    }

    /// `Modify::unwrap_connectivity_dim` (modify.rs:72-131), in place: GPU neighbour search with local ids under full PBC,
    /// `SearchConnectivity`'s adjacency in pair order, the reference's stack walk inside the plugin.  Returns the groups of
    /// LOCAL indices the reference returns as selections (`self.select(&sel_vec)`).
    pub fn unwrap_connectivity_dim(
        &self, coords: &mut [[f32; 3]], index: Option<&[usize]>, box9: &[f32; 9], cutoff: f32, dims: u8,
    ) -> Result<Vec<Vec<usize>>, EngineError> {
        check_index(index, coords.len(), "unwrap_connectivity_dim")?;
        let (ip, n) = idx_ptr(index);
        let nsel = index.map_or(coords.len(), |i| i.len());
        let mut off = vec![0u64; nsel + 1];
        let mut ids = vec![0u64; nsel.max(1)];
        let mut ng = 0usize;
        self.plugin.check(unsafe {
            (self.plugin.fns.unwrap_connectivity)(
                self.ctx, coords.as_mut_ptr() as *mut f32, coords.len(), ip, n, box9.as_ptr(), cutoff, dims, off.as_mut_ptr(),
                ids.as_mut_ptr(), &mut ng,
            )
        })?;
        Ok((0..ng).map(|g| ids[off[g] as usize..off[g + 1] as usize].iter().map(|&v| v as usize).collect()).collect())
    }

    /// `Measure::gyration` over a `ParSplit` (system.rs:193-213): selection k is `index[offsets[k]..offsets[k+1]]`.
    pub fn gyration_batch(
        &self, coords: &[[f32; 3]], index: &[usize], offsets: &[usize], masses: &[f32], box9: Option<&[f32; 9]>,
    ) -> Result<Vec<f32>, EngineError> {
        check_index(Some(index), coords.len(), "gyration_batch")?;
        check_offsets(offsets, index.len(), "gyration_batch")?;
        check_column(masses.len(), coords.len(), "gyration_batch: masses")?;
        let nsel = offsets.len().saturating_sub(1);
        let mut out = vec![0f32; nsel];
        self.plugin.check(unsafe {
            (self.plugin.fns.gyration_batch)(self.ctx, coords.as_ptr() as *const f32, coords.len(), index.as_ptr() as *const u64,
                                             offsets.as_ptr() as *const u64, nsel, masses.as_ptr(), box_ptr(box9), out.as_mut_ptr())
        })?;
        Ok(out)
    }
}

/// What one frame of a streamed fit returns: `fit_transform` (measure.rs:507-522) as (R column-major, t), and RMSD / centre of
/// mass / gyration of the fitted selection.
#[derive(Clone, Copy, Debug)]
pub struct FitRecord {
    pub r: [f32; 9],
    pub t: [f32; 3],
    pub rmsd: f32,
    pub com: [f32; 3],
    pub gyration: f32,
}

/// The per-frame fit loop of `benches/comparison_small.rs:14-25` for States in host memory (`analysis_task.rs:245-252` hands
/// them to the task one by one), at the rate the SELECTION crosses the link: `molar_hip_fit_stream_*`.  Up to three frames in
/// flight: `let t1 = fs.begin(&mut next, false)?; let rec = fs.end(t0)?;`.
pub struct FitStream<'e> {
    engine: &'e Engine,
    handle: *mut types::MolarHipFitStream,
    natoms: usize,
}

impl<'e> FitStream<'e> {
    /// `index` / `ref_index`: the selection in the frames and in the reference (None = all atoms); `masses`: the topology's
    /// column; `reference`: the frame fitted onto.
    pub fn new(
        engine: &'e Engine, natoms: usize, index: Option<&[usize]>, masses: &[f32], reference: &[[f32; 3]], ref_index: Option<&[usize]>,
        host_threads: i32,
    ) -> Result<Self, EngineError> {
        check_index(index, natoms, "FitStream::new (selection)")?;
        check_index(ref_index, reference.len(), "FitStream::new (reference)")?;
        check_column(masses.len(), natoms, "FitStream::new: masses")?;
        let (ip, n) = idx_ptr(index);
        let (rp, rn) = idx_ptr(ref_index);
        if index.is_some() != ref_index.is_some() || n != rn {
            return Err(EngineError::Sizes("FitStream::new: the two selections differ in size".into()));
        }
        let mut handle = std::ptr::null_mut();
        engine.plugin.check(unsafe {
            (engine.plugin.fns.fit_stream_create)(engine.ctx, natoms, ip, n, masses.as_ptr(), reference.as_ptr() as *const f32, reference.len(), rp,
                                                  host_threads, &mut handle)
        })?;
        Ok(FitStream { engine, handle, natoms })
    }

    /// Enqueues one frame.  With `apply` the fitted selection is written into `coords` by `end` (`Modify::apply_transform`,
    /// modify.rs:32-36): the borrow the ticket stands for must stay alive and untouched until then - hence `unsafe`.
    ///
    /// # Safety
    /// `coords` must outlive the matching `end` call and must not be read or written in between.
    pub unsafe fn begin(&mut self, coords: &mut [[f32; 3]], apply: bool) -> Result<i32, EngineError> {
        if coords.len() != self.natoms {
            return Err(EngineError::Sizes(format!("FitStream::begin: frame of {} atoms, stream built for {}", coords.len(), self.natoms)));
        }
        let mut ticket = -1i32;
        self.engine.plugin.check(unsafe { (self.engine.plugin.fns.fit_stream_begin)(self.handle, coords.as_mut_ptr() as *mut f32, apply as i32, &mut ticket) })?;
        Ok(ticket)
    }

    pub fn end(&mut self, ticket: i32) -> Result<FitRecord, EngineError> {
        let mut rec = FitRecord { r: [0.0; 9], t: [0.0; 3], rmsd: 0.0, com: [0.0; 3], gyration: 0.0 };
        self.engine.plugin.check(unsafe {
            (self.engine.plugin.fns.fit_stream_end)(self.handle, ticket, &mut rec.rmsd, rec.r.as_mut_ptr(), rec.t.as_mut_ptr(), rec.com.as_mut_ptr(),
                                                   &mut rec.gyration)
        })?;
        Ok(rec)
    }
}

impl Drop for FitStream<'_> {
    fn drop(&mut self) {
        unsafe { (self.engine.plugin.fns.fit_stream_destroy)(self.handle) }
    }
}

/// The remaining entry points (histogram-fused search, resident/pipelined search, batched fits, membrane smoothing,
/// lipid order, XTC reader, PeriodicBox helpers) are reachable through `Engine::raw()`; they take the same pointer/size
/// pairs and follow the count-then-fill convention documented in `include/molar_hip.h`.
impl Engine {
    pub fn raw(&self) -> (&MolarHipFns, *mut MolarHipCtx) {
        (&self.plugin.fns, self.ctx)
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Bilayer frames: `Membrane::compute` (molar_membrane/src/lib.rs:410-454) as one chained call per frame
// (molar_hip_membrane_frame_begin / _end / _fetch), the Rust side of `MembraneFrames` in include/molar_hip.hpp.

/// Index lists of a bilayer, as `Membrane::new` collects them per lipid (lib.rs:120-177): whole lipids, the three marker
/// sub-selections (head, mid, tail end) per lipid, and the tails' carbons with their bond orders.
pub struct MembraneLayout<'a> {
    pub natoms: usize,
    pub lipid_idx: &'a [usize],
    pub lipid_offsets: &'a [usize],
    pub marker_idx: &'a [usize],
    pub marker_offsets: &'a [usize],
    pub masses: &'a [f32],
    pub tail_idx: &'a [usize],
    pub tail_offsets: &'a [usize],
    pub tail_lipid: &'a [u32],
    pub tail_bonds: &'a [u8],
    pub cutoff: f32,
    pub order_type: i32,
    pub max_smooth_iter: i32,
    pub unwrap: bool,
    pub global_normal: Option<[f32; 3]>,
}

/// Per-lipid results of one frame copied to the host (the arrays `LipidGroup::frame_update` reads, lipid_group.rs).
pub struct MembraneFrame {
    pub valid: Vec<u8>,
    pub normals: Vec<[f32; 3]>,
    pub mean_curv: Vec<f32>,
    pub gauss_curv: Vec<f32>,
    pub area: Vec<f32>,
    pub nvert: Vec<u32>,
    pub neib_ids: Vec<u64>,
    pub patch_offsets: Vec<u64>,
    pub order: Vec<f32>,
}

/// Two frames in flight on one engine: `push(frame k+1)` returns the results of frame k.
pub struct MembraneFrames<'e> {
    engine: &'e Engine,
    plan: *mut MolarHipMembranePlan,
    nlipids: usize,
    natoms: usize,
    pending: Option<i32>,
}

impl<'e> MembraneFrames<'e> {
    pub fn new(engine: &'e Engine, l: &MembraneLayout) -> Result<Self, EngineError> {
        let k = l.lipid_offsets.len().saturating_sub(1);
        check_offsets(l.lipid_offsets, l.lipid_idx.len(), "MembraneFrames: lipids")?;
        check_offsets(l.marker_offsets, l.marker_idx.len(), "MembraneFrames: markers")?;
        check_offsets(l.tail_offsets, l.tail_idx.len(), "MembraneFrames: tails")?;
        check_index(Some(l.lipid_idx), l.natoms, "MembraneFrames: lipids")?;
        check_index(Some(l.marker_idx), l.natoms, "MembraneFrames: markers")?;
        check_index(Some(l.tail_idx), l.natoms, "MembraneFrames: tails")?;
        check_column(l.masses.len(), l.natoms, "MembraneFrames: masses")?;
        let ntails = l.tail_offsets.len().saturating_sub(1);
        if l.marker_offsets.len() != 3 * k + 1 || l.tail_lipid.len() != ntails || l.tail_bonds.len() + ntails < l.tail_idx.len() {
            return Err(EngineError::Sizes("MembraneFrames: three marker selections per lipid, one lipid and n-1 bond orders per tail".into()));
        }
        let d = MolarHipMembraneDesc {
            natoms: l.natoms,
            nlipids: k,
            lipid_idx: l.lipid_idx.as_ptr() as *const u64,
            lipid_offsets: l.lipid_offsets.as_ptr() as *const u64,
            marker_idx: l.marker_idx.as_ptr() as *const u64,
            marker_offsets: l.marker_offsets.as_ptr() as *const u64,
            masses: l.masses.as_ptr(),
            ntails,
            tail_idx: l.tail_idx.as_ptr() as *const u64,
            tail_offsets: l.tail_offsets.as_ptr() as *const u64,
            tail_lipid: l.tail_lipid.as_ptr(),
            tail_bonds: l.tail_bonds.as_ptr(),
            cutoff: l.cutoff,
            order_type: l.order_type,
            max_smooth_iter: l.max_smooth_iter,
            unwrap: l.unwrap as i32,
            use_global_normal: l.global_normal.is_some() as i32,
            global_normal: l.global_normal.unwrap_or([0.0; 3]),
        };
        let mut plan = std::ptr::null_mut();
        engine.plugin.check(unsafe { (engine.plugin.fns.membrane_plan_create)(engine.ctx, &d, &mut plan) })?;
        Ok(MembraneFrames { engine, plan, nlipids: k, natoms: l.natoms, pending: None })
    }

    /// `reset_valid_lipids` (lib.rs:269-273) with `None`, or the caller's flags; ends the frame in flight first.
    pub fn set_valid(&mut self, valid: Option<&[u8]>) -> Result<Option<MembraneFrame>, EngineError> {
        let last = self.finish()?;
        if let Some(v) = valid {
            if v.len() != self.nlipids {
                return Err(EngineError::Sizes(format!("set_valid: {} flags for {} lipids", v.len(), self.nlipids)));
            }
        }
        let p = valid.map_or(std::ptr::null(), |v| v.as_ptr());
        self.engine.plugin.check(unsafe { (self.engine.plugin.fns.membrane_plan_set_valid)(self.plan, p) })?;
        Ok(last)
    }

    /// Enqueue one frame (coordinates are unwrapped in place) and collect the frame pushed before it.
    pub fn push(&mut self, coords: &mut [[f32; 3]], box9: &[f32; 9]) -> Result<Option<MembraneFrame>, EngineError> {
        if coords.len() != self.natoms {
            return Err(EngineError::Sizes(format!("push: {} atoms, the layout was made for {}", coords.len(), self.natoms)));
        }
        let mut t = -1i32;
        self.engine.plugin.check(unsafe {
            (self.engine.plugin.fns.membrane_frame_begin)(self.plan, coords.as_mut_ptr() as *mut f32, box9.as_ptr(), &mut t)
        })?;
        let last = self.finish()?;
        self.pending = Some(t);
        Ok(last)
    }

    /// Collect the frame in flight, if any.
    pub fn finish(&mut self) -> Result<Option<MembraneFrame>, EngineError> {
        let Some(t) = self.pending.take() else { return Ok(None) };
        let f = &self.engine.plugin.fns;
        let mut v: MolarHipMembraneView = unsafe { std::mem::zeroed() };
        self.engine.plugin.check(unsafe { (f.membrane_frame_end)(self.plan, t, &mut v) })?;
        let k = self.nlipids;
        let mut r = MembraneFrame {
            valid: vec![0; k],
            normals: vec![[0.0; 3]; k],
            mean_curv: vec![0.0; k],
            gauss_curv: vec![0.0; k],
            area: vec![0.0; k],
            nvert: vec![0; k],
            neib_ids: vec![0; v.patch_entries + 4 * k],
            patch_offsets: vec![0; k + 1],
            order: vec![0.0; v.norder],
        };
        let mut o: MolarHipMembraneOut = unsafe { std::mem::zeroed() };
        o.valid = r.valid.as_mut_ptr();
        o.normals = r.normals.as_mut_ptr() as *mut f32;
        o.mean_curv = r.mean_curv.as_mut_ptr();
        o.gauss_curv = r.gauss_curv.as_mut_ptr();
        o.area = r.area.as_mut_ptr();
        o.nvert = r.nvert.as_mut_ptr();
        o.neib_ids = r.neib_ids.as_mut_ptr();
        o.patch_offsets = r.patch_offsets.as_mut_ptr();
        o.order = r.order.as_mut_ptr();
        self.engine.plugin.check(unsafe { (f.membrane_frame_fetch)(self.plan, t, &o) })?;
        Ok(Some(r))
    }
}

impl Drop for MembraneFrames<'_> {
    fn drop(&mut self) {
        unsafe { (self.engine.plugin.fns.membrane_plan_destroy)(self.plan) }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Frame-parallel driver for one node with several GPUs - the Rust side of `AnalysisTask::run_sharded` in
// include/molar_hip.hpp.  MolAR's `AnalysisTask::run` (analysis_task.rs:202-267) is a serial loop over frames whose
// body does not depend on earlier frames for the analyses on this path (search, fit/RMSD, Measure); frames therefore
// shard embarrassingly: one engine context and one task instance per device, contiguous blocks of frames dealt
// round-robin, no exchange until the end, then an integer reduction (`merge`) in worker order.

/// Number of GPUs the engine sees (`molar_hip_device_count`); 0 when the library or a GPU is missing.
pub fn device_count() -> usize {
    match Plugin::get_cached() {
        Ok(p) => unsafe { (p.fns.device_count)() }.max(0) as usize,
        Err(_) => 0,
    }
}

/// What a frame-parallel analysis provides.  `new` plays the role of `AnalysisTask::new`: every instance is built on the
/// FIRST consumed frame of the run (so a reference structure taken there is the same on every device); only the worker
/// that owns frame 0 also processes it.
pub trait ShardedTask: Send + Sized {
    type Frame: Send + Sync + Clone;
    fn new(engine: &Engine, first: &Self::Frame) -> Result<Self, EngineError>;
    /// `frame_index` is the position of the frame among the consumed frames of the whole run: per-frame results are
    /// tagged with it and put back in order by `merge`.
    fn process_frame(&mut self, engine: &Engine, frame_index: usize, frame: Self::Frame) -> Result<(), EngineError>;
    /// Folds another worker's instance into this one: integer accumulators (histogram bins, pair counts) add, per-frame
    /// series are concatenated and sorted by frame index.
    fn merge(&mut self, other: Self);
}

/// Runs `frames` (already windowed by -b/-e/--skip: the iterator is what `AnalysisTask::run` would consume, read by the
/// calling thread) through one worker per entry of `devices`, in blocks of `block` consecutive frames.  Returns the merged
/// instance and the number of frames consumed; `None` if the iterator was empty.
pub fn run_sharded<T, I>(devices: &[i32], block: usize, frames: I) -> Result<Option<(T, usize)>, EngineError>
where
    T: ShardedTask,
    I: Iterator<Item = T::Frame>,
{
    use std::sync::mpsc::sync_channel;
    if devices.is_empty() {
        return Err(EngineError::Sizes("run_sharded: no devices".into()));
    }
    let block = block.max(1);
    let mut frames = frames.enumerate().peekable();
    let first = match frames.peek() {
        Some((_, f)) => f.clone(),
        None => return Ok(None),
    };
    let first = &first;
    std::thread::scope(|scope| {
        let mut senders = Vec::with_capacity(devices.len());
        let mut handles = Vec::with_capacity(devices.len());
        for &dev in devices {
            let (tx, rx) = sync_channel::<(usize, T::Frame)>(2 * block + 2);
            senders.push(tx);
            handles.push(scope.spawn(move || -> Result<Option<(T, usize)>, EngineError> {
                let engine = Engine::new(dev)?;
                let mut task: Option<T> = None;
                let mut done = 0usize;
                for (index, frame) in rx {
                    if task.is_none() {
                        task = Some(T::new(&engine, first)?);
                    }
                    task.as_mut().unwrap().process_frame(&engine, index, frame)?;
                    done += 1;
                }
                Ok(task.map(|t| (t, done)))
            }));
        }
        for (index, frame) in frames {
            // a worker that failed has dropped its receiver: stop feeding, its error is collected below
            if senders[(index / block) % devices.len()].send((index, frame)).is_err() {
                break;
            }
        }
        drop(senders);
        let mut merged: Option<(T, usize)> = None;
        let mut failure = None;
        for h in handles {
            match h.join().unwrap_or_else(|_| Err(EngineError::Other("run_sharded: a worker panicked".into()))) {
                Ok(Some((t, n))) => match merged.as_mut() {
                    Some((head, total)) => {
                        head.merge(t);
                        *total += n;
                    }
                    None => merged = Some((t, n)),
                },
                Ok(None) => {}
                Err(e) => failure = failure.or(Some(e)),
            }
        }
        match failure {
            Some(e) => Err(e),
            None => Ok(merged),
        }
    })
}

/// The same with one reader PER WORKER (`run_sharded_own_readers` in include/molar_hip.hpp): `nframes` consumed frames are
/// known up front (an indexed trajectory: `FileHandler` random access, io.rs:691-760), worker `w` takes the contiguous block
/// `[w * ceil(n / W), ...)` and gets its frames from `open(first_index)` - its own reader, positioned at the block's first
/// consumed frame (with `--skip` the reader yields consumed frames only).  No producer thread, no channel: a single reader
/// decodes ~460 frames/s of 250k atoms per host thread, eight GPUs binning 4 k frames/s each would wait for it.  `first` is the
/// first consumed frame of the run, on which every instance is built.  Returns the merged instance and the frames consumed.
pub fn run_sharded_readers<T, O, R>(devices: &[i32], nframes: usize, first: &T::Frame, open: O) -> Result<Option<(T, usize)>, EngineError>
where
    T: ShardedTask,
    O: Fn(usize) -> Result<R, EngineError> + Sync,
    R: Iterator<Item = T::Frame>,
{
    if devices.is_empty() {
        return Err(EngineError::Sizes("run_sharded_readers: no devices".into()));
    }
    if nframes == 0 {
        return Ok(None);
    }
    let per = (nframes + devices.len() - 1) / devices.len();
    let open = &open;
    std::thread::scope(|scope| {
        let mut handles = Vec::with_capacity(devices.len());
        for (w, &dev) in devices.iter().enumerate() {
            let (lo, hi) = ((w * per).min(nframes), ((w + 1) * per).min(nframes));
            handles.push(scope.spawn(move || -> Result<Option<(T, usize)>, EngineError> {
                if lo == hi {
                    return Ok(None);
                }
                let engine = Engine::new(dev)?;
                let mut task = T::new(&engine, first)?;
                let mut done = 0usize;
                for (k, frame) in open(lo)?.take(hi - lo).enumerate() {
                    task.process_frame(&engine, lo + k, frame)?;
                    done += 1;
                }
                if done != hi - lo {
                    return Err(EngineError::Other(format!("run_sharded_readers: the reader of frames {lo}..{hi} ended after {done}")));
                }
                Ok(Some((task, done)))
            }));
        }
        let mut merged: Option<(T, usize)> = None;
        let mut failure = None;
        for h in handles {
            match h.join().unwrap_or_else(|_| Err(EngineError::Other("run_sharded_readers: a worker panicked".into()))) {
                Ok(Some((t, n))) => match merged.as_mut() {
                    Some((head, total)) => {
                        head.merge(t);
                        *total += n;
                    }
                    None => merged = Some((t, n)),
                },
                Ok(None) => {}
                Err(e) => failure = failure.or(Some(e)),
            }
        }
        match failure {
            Some(e) => Err(e),
            None => Ok(merged),
        }
    })
}
