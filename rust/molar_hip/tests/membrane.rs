//! Parity harness for the bilayer analysis: MolAR's own `molar_membrane::Membrane` against the committed fixture.
//!
//! SOURCE ONLY - the image this repository is built in has no Rust toolchain, so this file has never been compiled or run.
//! Why it exists: the reference holds no asserting test for any number `Membrane::compute` produces (its tests print), so the
//! CPU checker this repository's GPU path is compared with is, for the membrane, a careful reading pinned only against
//! independent geometry (qhull, least squares).  `tests/fixtures/membrane_cg/` holds a 200-lipid coarse-grained bilayer
//! (`bilayer.gro`, lipids at the edges split over the periodic boundary), the options (`options.toml`) and what the checker's
//! primitives, assembled in the reference's order, say `Membrane::new` + one `Membrane::compute` leave in every
//! `LipidMolecule` (`tests/golden/make_membrane_fixture.py`; `tests/test_rust_membrane_fixture_cpu.py` keeps the files equal
//! to a fresh computation and this file in step with the manifest).
//!
//!     cd rust/molar_hip && cargo test --test membrane          (dev-dependencies `molar`, `molar_membrane`; no GPU, no engine)
//!
//! Green = the membrane rows of the GPU parity tests are pinned by the reference.  Bars: lipid ids, patch lists with their
//! ORDER and validity exact; Voronoi neighbours as a ring (see `ring`); floats within 2e-5 (the tolerance the GPU path is
//! held to against the checker), areas within 1e-3 (the fan over doubled vertices).  Not compared: principal curvature directions (nalgebra's 2x2 eigenpair
//! order and sign are unspecified), vertex counts (below) and fields of lipids that are not valid.

use std::path::PathBuf;

use molar::prelude::*;
use molar_membrane::Membrane;

fn dir() -> PathBuf {
    [env!("CARGO_MANIFEST_DIR"), "tests", "fixtures", "membrane_cg"].iter().collect()
}
fn raw(key: &str) -> Vec<u8> {
    let p = dir().join(format!("{key}.bin"));
    std::fs::read(&p).unwrap_or_else(|e| panic!("{}: {e}", p.display()))
}
fn f32s(key: &str) -> Vec<f32> {
    raw(key).chunks_exact(4).map(|b| f32::from_le_bytes(b.try_into().unwrap())).collect()
}
fn u64s(key: &str) -> Vec<usize> {
    raw(key).chunks_exact(8).map(|b| u64::from_le_bytes(b.try_into().unwrap()) as usize).collect()
}
fn u32s(key: &str) -> Vec<u32> {
    raw(key).chunks_exact(4).map(|b| u32::from_le_bytes(b.try_into().unwrap())).collect()
}
/// Neighbours of a Voronoi cell as a canonical ring.  The clipping walk (molar/src/voronoi_cell.rs:107-205) leaves two
/// vertices on one neighbour's edge wherever a later bisector passes within rounding of an existing vertex, and where the ring
/// starts depends on the last cut: both flip with the last bit of the inputs.  Consecutive repeats are collapsed and the ring
/// is rotated to its smallest id; `exact` below counts the cells that agree vertex for vertex.
fn ring(ids: &[usize]) -> Vec<usize> {
    let n = ids.len();
    let mut out: Vec<usize> = (0..n).filter(|&k| n == 1 || ids[k] != ids[(k + n - 1) % n]).map(|k| ids[k]).collect();
    if out.is_empty() && n > 0 {
        out.push(ids[0]);
    }
    if let Some(k) = out.iter().enumerate().min_by_key(|(_, &v)| v).map(|(k, _)| k) {
        out.rotate_left(k);
    }
    out
}

fn close(got: Float, want: f32, what: &str) {
    let (g, w) = (got as f64, want as f64);
    assert!((g - w).abs() <= 2e-5 * w.abs().max(1.0), "{what}: {g} vs {w}");
}
fn close3(got: [Float; 3], want: &[f32], what: &str) {
    for d in 0..3 {
        close(got[d], want[d], &format!("{what}[{d}]"));
    }
}

#[test]
fn membrane_new_and_compute() {
    let mut sys = System::from_file(dir().join("bilayer.gro")).expect("fixture structure");
    let toml = std::fs::read_to_string(dir().join("options.toml")).expect("fixture options");
    // Membrane::new (molar_membrane/src/lib.rs:88-200): lipids split by residue in file order, each made whole, markers
    let mut memb = Membrane::new(&mut sys, &toml).expect("Membrane::new");
    let (head0, mid0, tail0) = (f32s("head_marker_new"), f32s("mid_marker_new"), f32s("tail_marker_new"));
    let k = head0.len() / 3;
    assert_eq!(memb.iter_all_lipids().count(), k, "number of lipids");
    for (i, lip) in memb.iter_all_lipids().enumerate() {
        assert_eq!(lip.id, i, "lipid ids follow the file");
        assert!(lip.valid);
        close3([lip.head_marker.x, lip.head_marker.y, lip.head_marker.z], &head0[3 * i..], &format!("lipid {i} head marker"));
        close3([lip.mid_marker.x, lip.mid_marker.y, lip.mid_marker.z], &mid0[3 * i..], &format!("lipid {i} mid marker"));
        close3([lip.tail_marker.x, lip.tail_marker.y, lip.tail_marker.z], &tail0[3 * i..], &format!("lipid {i} tail marker"));
    }

    // one frame (lib.rs:410-454): patches, initial normals, one smoothing pass, order
    memb.compute(&sys).expect("Membrane::compute");
    let valid = raw("valid");
    let (poff, pids) = (u64s("patch_offsets"), u64s("patch_ids"));
    let (noff, nids, nvert) = (u64s("neib_offsets"), u64s("neib_ids"), u32s("nvert"));
    let (head, normal) = (f32s("head_marker"), f32s("normal"));
    let (mean, gauss, area, order) = (f32s("mean_curv"), f32s("gaussian_curv"), f32s("area"), f32s("order"));
    let ntails = 2usize;
    let per_tail = order.len() / (k * ntails);
    let mut exact = 0usize;
    for (i, lip) in memb.iter_all_lipids().enumerate() {
        assert_eq!(lip.valid, valid[i] != 0, "lipid {i}: validity");
        assert_eq!(&lip.patch_ids[..], &pids[poff[i]..poff[i + 1]], "lipid {i}: patch ids (or their ORDER) differ");
        if !lip.valid {
            continue;
        }
        assert_eq!(ring(&lip.neib_ids), ring(&nids[noff[i]..noff[i + 1]]), "lipid {i}: Voronoi neighbours differ");
        exact += (lip.neib_ids[..] == nids[noff[i]..noff[i + 1]] && lip.voro_vertexes.len() == nvert[i] as usize) as usize;
        close3([lip.head_marker.x, lip.head_marker.y, lip.head_marker.z], &head[3 * i..], &format!("lipid {i} smoothed marker"));
        close3([lip.normal.x, lip.normal.y, lip.normal.z], &normal[3 * i..], &format!("lipid {i} normal"));
        close(lip.mean_curv, mean[i], &format!("lipid {i} mean curvature"));
        close(lip.gaussian_curv, gauss[i], &format!("lipid {i} gaussian curvature"));
        assert!(((lip.area - area[i] as Float) / area[i] as Float).abs() <= 1e-3, "lipid {i} area: {} vs {}", lip.area, area[i]);
        assert_eq!(lip.order.len(), ntails, "lipid {i}: tails");
        for t in 0..ntails {
            assert_eq!(lip.order[t].len(), per_tail, "lipid {i} tail {t}: order length");
            for c in 0..per_tail {
                close(lip.order[t][c], order[(i * ntails + t) * per_tail + c], &format!("lipid {i} tail {t} carbon {c} order"));
            }
        }
    }
    // identical arithmetic gives identical cells: if MolAR and the checker round alike, every cell agrees vertex for vertex
    println!("{exact} of {k} cells identical vertex for vertex");
    assert!(20 * exact >= 19 * k, "fewer than 95 % of the cells agree vertex for vertex");
}
