/*
 * molar_hip.h — C ABI of libmolar_hip.so, the MI355X (gfx950) engine for MolAR's per-frame
 * hot path: distance_search, PeriodicBox, Measure (COM/COG/gyration/inertia/rmsd/fit) and
 * Modify (apply_transform/unwrap_simple).
 *
 * This is the drop-in boundary.  It follows the reference's own runtime-plugin
 * convention (molar_gromacs/gromacs/wrapper.hpp:34-81): opaque handle + open/close,
 * thread-local last_error string, caller-owned output buffers filled after a count call,
 * plain C types only.  Each entry point names the reference interface it replaces
 * (paths relative to /root/reference/).
 *
 * Data layout handed over by MolAR (providers.rs:96-136, state.rs:22-28, atom_storage.rs:272):
 *   xyz   : whole frame, AoS float[3*natoms] (Vec<Pos>, Pos = Point3<f32>)
 *   idx   : sorted selection index slice (usize -> uint64_t), NULL = identity 0..natoms
 *   mass  : full-length column float[natoms], gathered through idx
 *   box9  : column-major float[9], COLUMNS are the box vectors a,b,c (periodic_box.rs:7-13)
 *   pbc   : PbcDims bit mask, bit d <=> dimension d (periodic_box.rs:81-114)
 *
 * Every pointer argument may be a HOST pointer or a DEVICE (HIP) pointer; the library detects
 * which (hipPointerGetAttributes) and stages host buffers through its own device buffers.
 * Device-resident inputs/outputs are used in place: no copy, kernels read/write them directly.
 *
 * Threading: a context is owned by one thread at a time (MolAR calls the search from one
 * thread, distance_search.rs:949); use one context per thread / per GPU.
 */
#ifndef MOLAR_HIP_H
#define MOLAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes: MeasureError (molar/src/measure.rs:732-762), PeriodicBoxError
 * (periodic_box.rs:131-144), LipidOrderError (measure.rs:718-729) */
enum {
    MOLAR_HIP_OK = 0,
    MOLAR_HIP_ERR_SIZES = 1,
    MOLAR_HIP_ERR_ZERO_MASS = 2,
    MOLAR_HIP_ERR_SVD = 3,           /* the rotation is found by Horn's quaternion method, which cannot fail to converge like an
                                      * SVD; this code is returned when the covariance is not finite (NaN / inf coordinates) */
    MOLAR_HIP_ERR_NO_PBC = 4,
    MOLAR_HIP_ERR_ZERO_LENGTH_VECTOR = 5,
    MOLAR_HIP_ERR_INVERSE_FAILED = 6,
    MOLAR_HIP_ERR_LIPID_TAIL_TOO_SHORT = 7,
    MOLAR_HIP_ERR_LIPID_NORMALS_COUNT = 8,
    MOLAR_HIP_ERR_LIPID_BOND_ORDER_COUNT = 9,
    MOLAR_HIP_ERR_ANGLE_TOO_SMALL = 10,
    MOLAR_HIP_ERR_INVALID_ARGUMENT = 50,
    MOLAR_HIP_ERR_TOO_LARGE = 51,
    MOLAR_HIP_ERR_NO_SEARCH = 52,
    MOLAR_HIP_ERR_IO = 53,        /* trajectory file: open/seek failure, truncated or corrupt frame */
    MOLAR_HIP_ERR_HIP = 100       /* HIP runtime failure, text in molar_hip_last_error() */
};

#define MOLAR_HIP_PBC_FULL 7u     /* periodic_box.rs:126 */
#define MOLAR_HIP_PBC_NONE 0u     /* periodic_box.rs:128 */

/* ------------------------------------------------------------------ lifecycle */

typedef struct molar_hip_ctx molar_hip_ctx;   /* opaque, like TprHandle (wrapper.hpp:34-36) */

/* Create an engine bound to HIP device `device`.  NULL on error (no GPU, bad ordinal);
 * message via molar_hip_last_error().  Mirrors tpr_open/tpr_close (wrapper.hpp:42-44). */
molar_hip_ctx *molar_hip_create(int device);
void molar_hip_destroy(molar_hip_ctx *ctx);
/* Thread-local text of the last failure on this thread; mirrors tpr_last_error (wrapper.hpp:46). */
const char *molar_hip_last_error(void);
const char *molar_hip_version(void);
/* Number of visible HIP devices, 0 if none / no driver (never fails). */
int molar_hip_device_count(void);
/* Run this context's kernels on an existing hipStream_t (e.g. torch's current stream). */
int molar_hip_set_stream(molar_hip_ctx *ctx, void *hip_stream);
int molar_hip_synchronize(molar_hip_ctx *ctx);

/* Per-kernel-class timing with HIP events recorded on the context's stream (what bench.py's
 * roofline line is computed from).  Classes: 0 grid build (bin/scan/scatter/place), 1 pair count
 * kernel, 2 offset scan, 3 pair fill kernel, 4 measure/fit kernels.  read() synchronises the
 * stream, returns the accumulated milliseconds and launch counts since the last read, and resets.
 * enable(ctx, 2): the resident searches record ONE span per search instead - class 5: count pass +
 * offset scan + fill pass between a single pair of events (an event record is a barrier packet of
 * 5-10 us on the stream, so the three bracketed passes of mode 1 read longer than they run). */
#define MOLAR_HIP_PROFILE_CLASSES 8
int molar_hip_profile_enable(molar_hip_ctx *ctx, int on);
int molar_hip_profile_read(molar_hip_ctx *ctx, float ms[MOLAR_HIP_PROFILE_CLASSES],
                           uint64_t launches[MOLAR_HIP_PROFILE_CLASSES]);

/* Device-memory ceilings of this GPU, measured in the caller's process (SURVEY.md §8d "measured device-copy ceiling
 * from a trivial float4 copy kernel in the same run").  copy: `bytes` device-to-device `reps` times, best of three
 * float4 kernels (grid-stride; four loads in flight per thread; the same non-temporal), (read + written bytes) / time.
 * write: a write-only float4 stream of `bytes` (the pair list is write traffic), bytes / time.  GB/s.
 * Diagnostics, not part of the reference. */
int molar_hip_copy_bandwidth(molar_hip_ctx *ctx, size_t bytes, int reps, float *gb_per_s);
int molar_hip_write_bandwidth(molar_hip_ctx *ctx, size_t bytes, int reps, float *gb_per_s);

/* ------------------------------------------------------------------ PeriodicBox (periodic_box.rs:15-23) */

typedef struct {
    float m[9];          /* column-major, columns a,b,c */
    float inv[9];        /* nalgebra try_inverse */
    int32_t nshift;      /* tric_corrections.len(), 0 for orthogonal boxes */
    float shifts[26 * 3];
} molar_hip_box;

/* PeriodicBox::from_matrix (periodic_box.rs:156-176): ERR_ZERO_LENGTH_VECTOR / ERR_INVERSE_FAILED. */
int molar_hip_box_from_matrix(const float m9[9], molar_hip_box *out);
/* PeriodicBox::from_vectors_angles (periodic_box.rs:188-235), angles in degrees. */
int molar_hip_box_from_vectors_angles(float a, float b, float c, float alpha, float beta, float gamma,
                                      molar_hip_box *out);
/* shortest_vector_dims (periodic_box.rs:286-318), host arithmetic identical to the device code. */
void molar_hip_box_shortest_vector(const molar_hip_box *box, const float v[3], uint8_t pbc, float out[3]);
/* get_lab_extents (periodic_box.rs:369-375): row sums, what sizes the search grid. */
void molar_hip_box_lab_extents(const molar_hip_box *box, float out[3]);
/* get_box_extents (:364-366): lengths of the box vectors. */
void molar_hip_box_extents(const molar_hip_box *box, float out[3]);
/* to_box_coords (:340-344) = inv*v, to_lab_coords (:356-360) = M*v. */
void molar_hip_box_to_box_coords(const molar_hip_box *box, const float v[3], float out[3]);
void molar_hip_box_to_lab_coords(const molar_hip_box *box, const float v[3], float out[3]);
/* is_inside (:348-352): all fractional coordinates in [0,1). */
int molar_hip_box_is_inside(const molar_hip_box *box, const float p[3]);
/* wrap_point / wrap_vec (:409-434), including the reference's `1.0 - bv` for negative fractions. */
void molar_hip_box_wrap_point(const molar_hip_box *box, const float p[3], float out[3]);

/* ------------------------------------------------------------------ distance search (distance_search.rs) */

enum {
    MOLAR_HIP_SEARCH_SINGLE = 0,      /* distance_search_single(_pbc)      :892-954 */
    MOLAR_HIP_SEARCH_DOUBLE = 1,      /* distance_search_double(_pbc)      :659-754 */
    MOLAR_HIP_SEARCH_WITHIN = 2,      /* distance_search_within(_pbc)      :519-598 */
    MOLAR_HIP_SEARCH_DOUBLE_VDW = 3   /* distance_search_double_vdw(_pbc)  :767-879 */
};

/* One search request.  Set 1 = (xyz1, natoms1, idx1, n1); set 2 likewise (unused for SINGLE).
 * The reference takes iterators of &Pos and of ids; here positions are gathered through idx
 * and the emitted ids are idx values (ids_local=0: iter_index(), molar_python/src/lib.rs:296-302)
 * or 0..n-1 (ids_local=1: modify.rs:78, all vdw drivers :791-792).
 * box9 == NULL selects the non-periodic driver, otherwise the *_pbc driver with `pbc`.
 * WITHIN without a box needs lower/upper (selection/ast.rs:597-612 computes them from min_max). */
typedef struct {
    int32_t kind;
    float cutoff;              /* ignored for DOUBLE_VDW (derived from the radii, :781-783) */
    const float *xyz1;
    size_t natoms1;
    const uint64_t *idx1;
    size_t n1;
    const float *xyz2;
    size_t natoms2;
    const uint64_t *idx2;
    size_t n2;
    const float *vdw1;         /* DOUBLE_VDW: radii per SELECTED atom, length n1 / n2 */
    const float *vdw2;
    int32_t ids_local;
    const float *box9;
    uint8_t pbc;
    const float *lower3;       /* WITHIN non-periodic only */
    const float *upper3;
} molar_hip_search_desc;

/* Phase 1 (count): builds the cell grid on the GPU, evaluates every cell pair of the
 * reference's search plan, returns the number of results in reference order.  The request
 * stays cached in ctx for the fill calls (tpr_n* then tpr_fill_*, wrapper.hpp:48-61). */
int molar_hip_search_count(molar_hip_ctx *ctx, const molar_hip_search_desc *desc, uint64_t *out_count);
/* Phase 2 (fill), results in exactly the reference's order (plan order, then i-major, j-minor;
 * distance_search.rs:949-953).  pairs: uint32 [count][2] (i,j); dist: float[count] = sqrt(d2)
 * (DistanceSearchOutput for (usize,usize,Float), :22-26).  Either may be NULL to skip it.
 * Outputs in DEVICE memory should be aligned to 16 bytes (pairs) and 8 bytes (dist) - the fill pass
 * writes two results per lane and store instruction (hipMalloc and framework allocators give 256
 * bytes).  An offset view that is not aligned like that is filled through the context's own buffers
 * and copied device to device: the same result, one more pass over it. */
int molar_hip_search_fill(molar_hip_ctx *ctx, uint32_t *pairs, float *dist);
/* Same, widened to MolAR's usize: separate i[], j[] arrays of uint64. */
int molar_hip_search_fill_usize(molar_hip_ctx *ctx, uint64_t *i, uint64_t *j, float *dist);
/* WITHIN results (DistanceSearchOutput for usize, :10-14): ids of set-1 atoms, duplicates kept
 * exactly as the reference emits them (callers sort+dedup, selection_expr.rs:112). */
int molar_hip_search_fill_ids(molar_hip_ctx *ctx, uint64_t *ids);
/* `within <cutoff> [pbc] of <inner>` as the SET its callers keep (LogicalNode::Within, selection/ast.rs:589-631): the raw
 * stream of distance_search_within(_pbc) (distance_search.rs:519-598, one id per plan entry in which the atom has a hit)
 * goes through SortedSet::from_unsorted (selection_expr.rs:112), so what reaches the user is the sorted, de-duplicated
 * ids of the first-set atoms with a second-set atom within the cutoff.  This pair computes exactly that set without
 * the stream: same grid, same plan entries (wrap / drop rules, wrapped entries through PeriodicBox::distance_squared
 * over the entry's dims), but an atom stops looking at its first hit in ANY entry, and the set comes out of a flag
 * array in ascending order.  `desc` must be of kind MOLAR_HIP_SEARCH_WITHIN (non-periodic: lower3 / upper3 as for the
 * stream form); ids are what the stream would carry (idx1 values, or 0..n1 with ids_local).  The `self` keyword
 * (:627-629) is the caller's union with the inner selection.  ids: uint64[count], host or device. */
int molar_hip_within_count(molar_hip_ctx *ctx, const molar_hip_search_desc *desc, uint64_t *out_count);
int molar_hip_within_fill(molar_hip_ctx *ctx, uint64_t *ids);
/* Consecutive `within` requests against one frame (selection/ast.rs:589-631 evaluated for several inner selections or cutoffs;
 * within_size_bench.rs:13-47) name the same first set: with the hold on, a molar_hip_within_count whose request has the same
 * first-set pointers and sizes, the same box and periodicity and comes to the same grid reuses the staged coordinates and the
 * grid of the request before it.  The caller promises that those coordinates do not change while the hold is on (for a Rust
 * caller: while it holds the `&State`); any other search on the context, or on = 0, ends the reuse.  A first set in host
 * memory is additionally checked by a fingerprint of the array (both ends and 512 atoms spread over it), so a frame updated
 * in place - or a new array at the old address - is staged again; a set in device memory is read in place, and there the
 * promise is all there is: toggle the hold per frame. */
int molar_hip_within_hold(molar_hip_ctx *ctx, int on);
/* SearchConnectivity (molar/src/connectivity.rs:8-60: `for (i, j) in pairs { conn[i].push(j); conn[j].push(i) }`) of a
 * single-selection search, built on the device from the resident pair list.  CSR over the request's id range - local ids
 * (desc->ids_local): the selection's length; global ids: natoms - with every list in the reference's push order; an atom
 * without a pair has an empty list (the reference's map has no key for it).  Count-then-fill like the searches: the first
 * call runs the search (the request must be of kind MOLAR_HIP_SEARCH_SINGLE), builds the CSR in context-owned device memory
 * and returns rows and entries (= 2 x pairs, < 2^31: MOLAR_HIP_ERR_TOO_LARGE beyond); the second copies offsets[rows + 1] and
 * neigh[entries] to host or device memory. */
int molar_hip_search_connectivity(molar_hip_ctx *ctx, const molar_hip_search_desc *desc, uint64_t *out_rows, uint64_t *out_entries);
int molar_hip_search_connectivity_fill(molar_hip_ctx *ctx, uint64_t *offsets, uint64_t *neigh);
/* Modify::unwrap_connectivity_dim (molar/src/modify.rs:72-131): neighbour search of the selection with local ids under
 * full PBC on the GPU, SearchConnectivity's adjacency in pair order (connectivity.rs:19-35) and the reference's stack
 * walk on the host - every atom is moved to the closest image (over `pbc_dims`) of the atom it was reached from.  xyz:
 * float[natoms][3], host or device, modified in place.  Returns the selections of the reference's result as a CSR of
 * LOCAL indices, each sorted (select(&sel_vec)); as there, the atom a connected component starts from is not a member
 * of its group and one-atom components give no group.  group_offsets: capacity n + 1 (or natoms + 1 with idx == NULL),
 * group_ids: capacity n; either may be NULL (the coordinates are unwrapped all the same, *ngroups is still counted).
 * Errors: MOLAR_HIP_ERR_NO_PBC without a box. */
int molar_hip_unwrap_connectivity(molar_hip_ctx *ctx, float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                                  const float *box9, float cutoff, uint8_t pbc_dims, uint64_t *group_offsets,
                                  uint64_t *group_ids, size_t *ngroups);
/* Grid dims of the cached search (Grid::get_dims, :212-214). */
int molar_hip_search_grid_dims(molar_hip_ctx *ctx, uint64_t dims[3]);
/* Which pair kernels the context's last search of a fixed-cutoff kind ran: *lanes = 16 or 32 for the small-cell kernels (frames of
 * a few atoms per cell: contact / hydrogen-bond cutoffs), 0 for the regular ones, -1 / -2 for the instances that keep second cells
 * of up to 1024 / 2048 atoms in registers (frames of more than 448 / 1000 atoms per cell).  The choice follows the atoms per OCCUPIED cell
 * once a search of the same shape (kind, set sizes, grid dims) has finished on this context - every grid build counts its occupied
 * cells and the host learns the count with the result sizes - and the atoms per cell until then: a slab or a solute in a mostly
 * empty periodic box has few atoms per cell on average and many per occupied cell, and moves to the regular kernels from its
 * second or third frame on.  occupied_cells (may be NULL): the counts that search went by, 0 = not known then.  Diagnostic; the
 * results do not depend on the choice. */
int molar_hip_search_cell_kernels(molar_hip_ctx *ctx, int32_t *lanes, uint64_t occupied_cells[2]);
/* Device-resident result of the cached search: fills ctx-owned buffers (reused across frames)
 * and returns their device addresses; valid until the next search on this ctx. */
int molar_hip_search_fill_device(molar_hip_ctx *ctx, const uint32_t **d_pairs, const float **d_dist);
/* The resident path in one call and ONE host round trip: count, offset scan and fill are enqueued back to back
 * into ctx-owned buffers sized by earlier frames of the trajectory; if a buffer turns out too small (first
 * frame) it grows and the affected pass repeats.  Same result, order and validity rules as
 * molar_hip_search_count + molar_hip_search_fill_device; the cached search stays available to the other fill
 * variants.  The result is complete in device memory when the call returns.  Not for WITHIN (ids, not pairs). */
int molar_hip_search_resident(molar_hip_ctx *ctx, const molar_hip_search_desc *desc, uint64_t *out_count,
                              const uint32_t **d_pairs, const float **d_dist);
/* Which planes the resident searches (molar_hip_search_resident, _begin / _end) fill: want_dist = 0 selects the
 * DistanceSearchOutput of (usize, usize) (distance_search.rs:14-20) - (i, j) only, 8 bytes per result, no square roots - for
 * consumers that never read the distances (SearchConnectivity, patches); the d_dist pointers then come back NULL.  Not while
 * pipelined searches are in flight.  Default: both planes. */
int molar_hip_search_resident_planes(molar_hip_ctx *ctx, int want_dist);
/* The same search split in two so that a per-frame loop never leaves the GPU idle: _begin enqueues everything for
 * one frame and returns at once with a ticket (0 or 1); _end waits for that frame only and returns its result.
 * Two searches may be in flight, each with its own result set, so the loop is
 *     begin(frame k+1); end(frame k); consume k; ...
 * and the host work of frame k+1 (and the result round trip of frame k) hides behind the kernels of frame k.
 * The kernels still run in order on the context's one stream.  A frame that outgrows a buffer is repeated inside
 * _end (after the younger search has drained), so `desc` and everything it points to must stay valid until _end.
 * A result set is valid until the second _begin after the one that produced it (ticket 0's set is also the one
 * molar_hip_search_resident and molar_hip_search_fill_device write: end the tickets before mixing those in).
 * On a context that owns its stream, with inputs already in device memory, the grid of the new frame is built on an
 * internal side stream while the kernels of the frame in flight run; the inputs must therefore be complete in
 * memory when _begin is called (a context created on a caller's stream keeps everything on that stream).  Errors: both sets in flight
 * (_begin), unknown or already finished ticket (_end). */
int molar_hip_search_resident_begin(molar_hip_ctx *ctx, const molar_hip_search_desc *desc, int32_t *ticket);
int molar_hip_search_resident_end(molar_hip_ctx *ctx, int32_t ticket, uint64_t *out_count, const uint32_t **d_pairs,
                                  const float **d_dist);
/* Consumer-fused variant: never materialises pairs; every emitted distance d goes through
 * Histogram1D::add_one (molar_membrane/src/stats.rs:29-35): b=floor(n*(d-min)/(max-min)),
 * counted in integers (bins: uint64[nbins], accumulated INTO, so frames can be summed).  With `bins` in device
 * memory the sum stays on the GPU, and with out_count == NULL the call returns without waiting for the kernels
 * (frames of a trajectory queue up back to back; molar_hip_synchronize before reading the bins).  Calls of that
 * asynchronous form on a context that owns its stream build their grid on the internal side stream under the
 * histogram kernel of the frame before: as for _begin, the inputs must be complete in memory at the call. */
int molar_hip_search_histogram(molar_hip_ctx *ctx, const molar_hip_search_desc *desc, float hmin,
                               float hmax, size_t nbins, uint64_t *bins, uint64_t *out_count);
/* The same for a block of a trajectory: `nframes` frames of the request's first set, frame k at desc->xyz1 + k * xyz1_stride
 * floats (second set: xyz2 + k * xyz2_stride), box of frame k at boxes9 + 9 k (NULL: desc->box9 for every frame) - the state
 * iterator of analysis_task.rs:245-252 / io.rs:198-271 handed over a window at a time.  The sums in `bins` are those of
 * nframes calls of molar_hip_search_histogram (integer bins do not care how their pairs are grouped).  Periodic requests of
 * kind SINGLE or DOUBLE whose coordinates, indices and bins are all in device memory, on a context that owns its stream, run in
 * groups of up to sixteen frames that share their launches - the grids of a group are built together on the side stream while the group before
 * is still in its histogram kernel, one plan launch and one persistent kernel walk the slots of all of them - and the call
 * does not wait (molar_hip_synchronize before reading the bins; the frames must be complete in memory at the call).  Anything
 * else is walked frame by frame through molar_hip_search_histogram. */
int molar_hip_search_histogram_frames(molar_hip_ctx *ctx, const molar_hip_search_desc *desc, size_t nframes, size_t xyz1_stride,
                                      size_t xyz2_stride, const float *boxes9, float hmin, float hmax, size_t nbins,
                                      uint64_t *bins);
/* Host arithmetic, no GPU: the table the fused histogram bins with.  Histogram1D::add_one's bin (stats.rs:29-35) is a
 * non-decreasing function of the squared distance; edges[b], b = 0..nbins (nbins + 1 floats), is the smallest
 * non-negative float d2 whose bin floor(n*(sqrt(d2)-min)/(max-min)) is >= b, found by bisection with the formula
 * itself, so "largest b with edges[b] <= d2" IS the formula.  INVALID_ARGUMENT unless min < max, both finite. */
int molar_hip_histogram_edges(float hmin, float hmax, size_t nbins, float *edges);

/* ---- the drivers for MolAR built with its `f64` feature (Float = f64, aliases.rs:10-13): every operation in double -
 * cell assignment, the predicate d2 <= cutoff^2 and the distances - results as (usize, usize, f64) columns or usize ids.
 * Same request / count-then-fill convention as above.  Coordinates, index lists, radii and the result columns may be host or
 * device memory (device memory is used in place); grid and plan are built on the device.  The matrix-core count and the
 * pipelined resident forms exist for f32 only. */
typedef struct {
    int32_t kind;
    double cutoff;
    const double *xyz1;
    size_t natoms1;
    const uint64_t *idx1;
    size_t n1;
    const double *xyz2;
    size_t natoms2;
    const uint64_t *idx2;
    size_t n2;
    const double *vdw1;
    const double *vdw2;
    int32_t ids_local;
    const double *box9;
    uint8_t pbc;
    const double *lower3;
    const double *upper3;
} molar_hip_search_desc_f64;
int molar_hip_search_count_f64(molar_hip_ctx *ctx, const molar_hip_search_desc_f64 *desc, uint64_t *out_count);
int molar_hip_search_fill_f64(molar_hip_ctx *ctx, uint64_t *i, uint64_t *j, double *dist);
int molar_hip_search_fill_ids_f64(molar_hip_ctx *ctx, uint64_t *ids);
int molar_hip_search_grid_dims_f64(molar_hip_ctx *ctx, uint64_t dims[3]);

/* ------------------------------------------------------------------ Measure (measure.rs) */

/* min_max :22-36 */
int molar_hip_min_max(molar_hip_ctx *ctx, const float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                      float lower[3], float upper[3]);
/* center_of_geometry :39-47 */
int molar_hip_center_of_geometry(molar_hip_ctx *ctx, const float *xyz, size_t natoms, const uint64_t *idx,
                                 size_t n, float out[3]);
/* center_of_mass :60-75 (ERR_ZERO_MASS) */
int molar_hip_center_of_mass(molar_hip_ctx *ctx, const float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                             const float *mass, float out[3]);
/* center_of_geometry_pbc_dims :156-168 / center_of_mass_pbc_dims :197-220 (ERR_NO_PBC if box9==NULL) */
int molar_hip_center_of_geometry_pbc(molar_hip_ctx *ctx, const float *xyz, size_t natoms, const uint64_t *idx,
                                     size_t n, const float *box9, uint8_t pbc, float out[3]);
int molar_hip_center_of_mass_pbc(molar_hip_ctx *ctx, const float *xyz, size_t natoms, const uint64_t *idx,
                                 size_t n, const float *mass, const float *box9, uint8_t pbc, float out[3]);
/* gyration :78-87, gyration_pbc :222-232 (box9 != NULL) */
int molar_hip_gyration(molar_hip_ctx *ctx, const float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                       const float *mass, const float *box9, float *out);
/* inertia :90-99 / inertia_pbc :234-244: moments ascending, axes column-major (cols = axes);
 * tensor9 (optional) receives the raw tensor (column-major). */
int molar_hip_inertia(molar_hip_ctx *ctx, const float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                      const float *mass, const float *box9, float moments[3], float axes9[9], float tensor9[9]);
/* rmsd :485-504 (ERR_SIZES), rmsd_mw :538-558 (masses of selection 1) */
int molar_hip_rmsd(molar_hip_ctx *ctx, const float *xyz1, size_t natoms1, const uint64_t *idx1, size_t n1,
                   const float *xyz2, size_t natoms2, const uint64_t *idx2, size_t n2, float *out);
int molar_hip_rmsd_mw(molar_hip_ctx *ctx, const float *xyz1, size_t natoms1, const uint64_t *idx1, size_t n1,
                      const float *mass1, const float *xyz2, size_t natoms2, const uint64_t *idx2, size_t n2,
                      float *out);
/* fit_transform :507-522 / fit_transform_at_origin :525-535 (at_origin != 0):
 * IsometryMatrix3 as R (column-major) and t, p -> R p + t. */
int molar_hip_fit_transform(molar_hip_ctx *ctx, const float *xyz1, size_t natoms1, const uint64_t *idx1,
                            size_t n1, const float *mass1, const float *xyz2, size_t natoms2,
                            const uint64_t *idx2, size_t n2, const float *mass2, int at_origin, float R9[9],
                            float t3[3]);

/* ---- MolAR built with its `f64` feature (Float = f64: molar/src/aliases.rs:10-13, molar/Cargo.toml:56-60).
 * The Measure / Modify methods (centres, gyration, rmsd, fit, inertia, min_max, apply, translate, periodic centres and unwrap) on double-precision coordinates and masses, same argument meaning and
 * error codes as the f32 entries above; every per-atom term is formed and accumulated in f64 (two passes where the
 * reference has two: centre, then centred terms).  The search is f32 only. */
/* center_of_geometry :39-47, center_of_mass :60-75 (ERR_ZERO_MASS) */
int molar_hip_center_of_geometry_f64(molar_hip_ctx *ctx, const double *xyz, size_t natoms, const uint64_t *idx,
                                     size_t n, double out[3]);
int molar_hip_center_of_mass_f64(molar_hip_ctx *ctx, const double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                                 const double *mass, double out[3]);
/* gyration :78-87 */
int molar_hip_gyration_f64(molar_hip_ctx *ctx, const double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                           const double *mass, double *out);
/* rmsd :485-504 (ERR_SIZES), rmsd_mw :538-558 */
int molar_hip_rmsd_f64(molar_hip_ctx *ctx, const double *xyz1, size_t natoms1, const uint64_t *idx1, size_t n1,
                       const double *xyz2, size_t natoms2, const uint64_t *idx2, size_t n2, double *out);
int molar_hip_rmsd_mw_f64(molar_hip_ctx *ctx, const double *xyz1, size_t natoms1, const uint64_t *idx1, size_t n1,
                          const double *mass1, const double *xyz2, size_t natoms2, const uint64_t *idx2, size_t n2,
                          double *out);
/* fit_transform :507-522 / fit_transform_at_origin :525-535: R column-major, p -> R p + t */
int molar_hip_fit_transform_f64(molar_hip_ctx *ctx, const double *xyz1, size_t natoms1, const uint64_t *idx1,
                                size_t n1, const double *mass1, const double *xyz2, size_t natoms2,
                                const uint64_t *idx2, size_t n2, const double *mass2, int at_origin, double R9[9],
                                double t3[3]);
/* min_max :22-36; inertia :90-99 (moments ascending, axes column-major, optional raw tensor); translate modify.rs:16-23 */
int molar_hip_min_max_f64(molar_hip_ctx *ctx, const double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                          double lower[3], double upper[3]);
int molar_hip_inertia_f64(molar_hip_ctx *ctx, const double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                          const double *mass, double moments[3], double axes9[9], double tensor9[9]);
int molar_hip_translate_f64(molar_hip_ctx *ctx, double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                            const double shift3[3]);
/* the per-frame loop of benches/comparison_small.rs:14-25 in f64, same argument meaning as molar_hip_fit_rmsd_batch:
 * every frame's selection fitted onto the reference selection (masses of the frame's atoms; the reference centre with
 * the same column through ref_idx), RMSD / centre of mass / gyration of the FITTED selection, frames moved if apply.
 * Outputs (each optional): rmsd[F], R[F][9] column-major, t[F][3], com[F][3], gyr[F]. */
int molar_hip_fit_rmsd_batch_f64(molar_hip_ctx *ctx, double *frames, size_t nframes, size_t natoms,
                                 const uint64_t *idx, size_t n, const double *mass, const double *ref_xyz,
                                 size_t ref_natoms, const uint64_t *ref_idx, int apply, double *rmsd_out,
                                 double *R_out, double *t_out, double *com_out, double *gyr_out);
/* the periodic centres (:156-168, :197-220), gyration_pbc (:222-232) and unwrap_simple_dim (modify.rs:40-54) with an f64
 * PeriodicBox built from box9 (column-major, columns a,b,c) exactly as PeriodicBox::from_matrix does: ERR_NO_PBC if
 * box9 == NULL, ERR_ZERO_LENGTH_VECTOR / ERR_INVERSE_FAILED from the construction. */
int molar_hip_center_of_geometry_pbc_f64(molar_hip_ctx *ctx, const double *xyz, size_t natoms, const uint64_t *idx,
                                         size_t n, const double *box9, uint8_t pbc, double out[3]);
int molar_hip_center_of_mass_pbc_f64(molar_hip_ctx *ctx, const double *xyz, size_t natoms, const uint64_t *idx,
                                     size_t n, const double *mass, const double *box9, uint8_t pbc, double out[3]);
int molar_hip_gyration_pbc_f64(molar_hip_ctx *ctx, const double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                               const double *mass, const double *box9, double *out);
int molar_hip_unwrap_simple_f64(molar_hip_ctx *ctx, double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                                const double *box9, uint8_t pbc);
/* rotate (modify.rs:25-30) about the origin, in place; principal_transform :102-109 / _pbc :246-257 (box9 != NULL) */
int molar_hip_rotate_f64(molar_hip_ctx *ctx, double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                         const double unit_axis3[3], double angle);
int molar_hip_principal_transform_f64(molar_hip_ctx *ctx, const double *xyz, size_t natoms, const uint64_t *idx,
                                      size_t n, const double *mass, const double *box9, double R9[9], double t3[3]);
/* Measure::lipid_tail_order (measure.rs:270-422) in f64, CSR layout and error codes of molar_hip_lipid_tail_order */
int molar_hip_lipid_tail_order_f64(molar_hip_ctx *ctx, const double *xyz, size_t natoms, const uint64_t *idx,
                                   const uint64_t *tail_offsets, size_t ntails, int order_type, const double *normals,
                                   const uint64_t *normal_offsets, const uint8_t *bond_orders, double *out);
/* inertia_pbc :234-244 */
int molar_hip_inertia_pbc_f64(molar_hip_ctx *ctx, const double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                              const double *mass, const double *box9, double moments[3], double axes9[9],
                              double tensor9[9]);
/* apply_transform (modify.rs:32-36), in place */
int molar_hip_apply_transform_f64(molar_hip_ctx *ctx, double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                                  const double R9[9], const double t3[3]);

/* Batched selections (CSR): selection k = idx[offsets[k] .. offsets[k+1]).  Replaces the rayon loops over
 * ParSplit sub-selections / per-lipid selections (selection/system.rs:193-213, molar_membrane/src/lib.rs:
 * 135-137, lipid_molecule.rs:65-99) where one GPU launch per ~100-atom selection would be slower than the
 * serial loop.  out: float[K][3].  mass == NULL => center_of_geometry, else center_of_mass (ERR_ZERO_MASS if
 * any selection has zero total mass). */
int molar_hip_center_batch(molar_hip_ctx *ctx, const float *xyz, size_t natoms, const uint64_t *idx,
                           const uint64_t *offsets, size_t nsel, const float *mass, float *out);
/* Modify::unwrap_simple_dim (modify.rs:40-54) applied to every selection of the CSR list, in place. */
int molar_hip_unwrap_simple_batch(molar_hip_ctx *ctx, float *xyz, size_t natoms, const uint64_t *idx,
                                  const uint64_t *offsets, size_t nsel, const float *box9, uint8_t pbc);
/* Membrane::compute_initial_normals (molar_membrane/src/lib.rs:456-505), host arithmetic: tail->head unit
 * vectors, then two neighbour-averaging passes over the patches (CSR patch_offsets/patch_ids; the second
 * pass updates normals in place, in lipid order, exactly like the reference loop).  valid may be NULL. */
int molar_hip_membrane_initial_normals(size_t nlipids, const float *head_markers, const float *tail_markers,
                                       const uint64_t *patch_offsets, const uint64_t *patch_ids,
                                       const uint8_t *valid, float *normals_out);

/* Membrane::compute_patches' list building (molar_membrane/src/lib.rs:548-557) from the (i, j) pairs of the
 * marker search, in pair order: patch_ids[i].push(j); patch_ids[j].push(i).  Host arrays; patch_offsets[K+1],
 * patch_ids[2*npairs]. */
int molar_hip_membrane_patches_from_pairs(const uint32_t *pairs, size_t npairs, size_t nlipids,
                                          uint64_t *patch_offsets, uint64_t *patch_ids);
/* One iteration of Membrane::smooth (molar_membrane/src/lib.rs:661-812) for all lipids on the GPU: local
 * frame from the lipid's normal (lipid_molecule.rs:190-196), patch markers into that frame through
 * PeriodicBox::shortest_vector, quadric fit (get_quad_coefs :844-863), Voronoi cell of the marker among its patch
 * (molar/src/voronoi_cell.rs:62-211), curvatures + fitted normal (lipid_molecule.rs:102-188), cell area, fitted
 * patch points, then the scatter-average of the markers (:781-809).  Replaces the rayon par_iter over lipids.
 * Host pointers.  Patches are CSR over lipid ids (compute_patches :539-558).  The state arrays are IN/OUT exactly
 * like the fields of LipidMolecule: a lipid that is (or becomes) invalid keeps its previous values; optional
 * outputs may be NULL.  Lipid i owns slots [patch_offsets[i] + 4*i, patch_offsets[i+1] + 4*(i+1)) of neib_ids
 * and voro_vertexes and fills the first nvert[i] of them (a cell has at most patch_len + 4 vertices).
 * Eigenpairs: nalgebra's symmetric_eigen leaves order and sign open; here princ_curvs are descending and each
 * direction has a positive first non-zero local component. */
typedef struct {
    size_t nlipids;
    const uint64_t *patch_offsets;   /* [nlipids+1] */
    const uint64_t *patch_ids;       /* [patch_offsets[nlipids]] */
} molar_hip_membrane_patches;
typedef struct {
    float *head_markers;             /* [K][3]  required */
    float *normals;                  /* [K][3]  required: in = normal defining the local frame, out = fitted normal */
    uint8_t *valid;                  /* [K]     required */
    float *quad_coefs;               /* [K][6]  a,b,c,d,e,f of z = a x^2 + b y^2 + c xy + d x + e y + f */
    float *mean_curv, *gauss_curv;   /* [K] */
    float *princ_curvs;              /* [K][2] */
    float *princ_dirs;               /* [K][2][3] */
    float *area;                     /* [K] */
    uint32_t *nvert;                 /* [K] */
    uint64_t *neib_ids;              /* [E + 4K] slotted, see above */
    float *voro_vertexes;            /* [E + 4K][3] slotted */
    float *fitted_patch_points;      /* [E][3] aligned with patch_ids */
} molar_hip_membrane_state;
int molar_hip_membrane_smooth(molar_hip_ctx *ctx, const molar_hip_membrane_patches *patches, const float *box9,
                              molar_hip_membrane_state *state);

/* patches_from_nth_shell (molar_membrane/src/lib.rs:562-583), host arithmetic: from the Voronoi neighbours a smoothing
 * pass left (nvert / neib_ids of molar_hip_membrane_state, slotted by patch_offsets), the patch of every valid lipid
 * becomes its n-th neighbour shell - the direct neighbours widened (n_shells - 2) times by the neighbours of every
 * member, the lipid itself included from n_shells = 3 on as in the reference; lipids that are not valid keep the patch
 * they have.  The reference collects a HashSet (order unspecified); here ids ascend.  Count-then-fill: with out_ids ==
 * NULL (or too small a capacity) only out_offsets[nlipids + 1] and *needed are written.  nvert[i] of a VALID lipid above
 * its slots (patch length + 4) is an error; a lipid that is not valid may still be walked as a member of a shell (it
 * keeps the neighbours an earlier pass left, as in the reference) and its count is clamped to its slots. */
int molar_hip_membrane_nth_shell_patches(size_t nlipids, const uint8_t *valid, const uint64_t *patch_offsets,
                                         const uint64_t *patch_ids, const uint32_t *nvert, const uint64_t *neib_ids,
                                         size_t n_shells, uint64_t *out_offsets, uint64_t *out_ids, size_t capacity,
                                         size_t *needed);
/* smooth_curvature (lib.rs:584-621), host arithmetic, in place: mean and Gaussian curvature of every valid lipid
 * averaged with those of the valid members of its n-th neighbour shell (sums in ascending id, f32, from the values
 * before the call); n_shells == 0 leaves everything as it is. */
int molar_hip_membrane_smooth_curvature(size_t nlipids, const uint8_t *valid, const uint64_t *patch_offsets,
                                        const uint32_t *nvert, const uint64_t *neib_ids, size_t n_shells,
                                        float *mean_curv, float *gauss_curv);

/* One whole frame of Membrane::compute (molar_membrane/src/lib.rs:410-454) as a chain of kernels with no host round
 * trip inside: unwrap of every lipid (lipid_molecule.rs:75-76) -> head / mid / tail-end markers (:65-99) -> PBC search
 * among the valid lipids' head markers and the patch lists in push order (compute_patches, lib.rs:539-558) ->
 * compute_initial_normals (:456-505, including the second pass that updates normals in place in lipid order) ->
 * max_smooth_iter iterations of smooth (:661-812) -> lipid_tail_order of every tail with its lipid's normal (:435-443).
 * The stages are the ones behind molar_hip_unwrap_simple_batch, _center_batch, _search_*, _membrane_patches_from_pairs,
 * _membrane_initial_normals, _membrane_smooth and _lipid_tail_order, and give the same bits; what differs is that the
 * pair list, the patches and every per-lipid array stay in device memory between them.
 *
 * A plan holds what is constant over a trajectory (the index lists, masses, options) and the per-lipid `valid` flags,
 * which the reference carries from frame to frame (LipidMolecule::valid: a lipid dropped by smooth stays out of the
 * patches of later frames until reset_valid_lipids, lib.rs:269-273).  Two frames may be in flight:
 *     begin(frame k+1); end(frame k); fetch / use the device view of k; ...
 * so the host work of a frame hides behind the kernels of the one before it.  Frames are chained on the context's
 * stream in begin order, `valid` included.  The result of a frame stays readable until the second _begin after its own.
 * Sizes that vary with the frame (pairs, patch entries) are provisioned from earlier frames; a frame that outgrows
 * them is repeated inside _end, transparently.  Options outside this entry (patches from the n-th neighbour shell,
 * curvature smoothing over shells: lib.rs:562-621) stay with the staged calls.  All pointers of the description are
 * host memory and are copied at creation. */
typedef struct molar_hip_membrane_plan molar_hip_membrane_plan;
typedef struct {
    size_t natoms, nlipids;
    const uint64_t *lipid_idx;        /* CSR of whole lipids: lipid k = lipid_idx[lipid_offsets[k] .. lipid_offsets[k+1]) */
    const uint64_t *lipid_offsets;    /* [nlipids + 1] */
    const uint64_t *marker_idx;       /* CSR of 3 * nlipids selections: head, mid, tail-end of lipid 0, head of lipid 1, ... */
    const uint64_t *marker_offsets;   /* [3 * nlipids + 1] */
    const float *masses;              /* [natoms] */
    size_t ntails;
    const uint64_t *tail_idx;         /* CSR of the tails' carbons in chain order, layout of molar_hip_lipid_tail_order */
    const uint64_t *tail_offsets;     /* [ntails + 1] */
    const uint32_t *tail_lipid;       /* [ntails] the lipid whose normal a tail is measured against */
    const uint8_t *tail_bonds;        /* bond orders, layout of molar_hip_lipid_tail_order */
    float cutoff;                     /* patch search radius (MembraneOptions::cutoff) */
    int32_t order_type;               /* 0 Sz, 1 Scd, 2 ScdCorr */
    int32_t max_smooth_iter;          /* >= 1 */
    int32_t unwrap;                   /* make every lipid whole before the markers are taken */
    int32_t use_global_normal;        /* order against global_normal instead of the lipid's normal */
    float global_normal[3];
} molar_hip_membrane_desc;
/* Device addresses of one frame's results (valid as described above); E = patch entries of the frame. */
typedef struct {
    size_t nlipids, patch_entries, npairs;
    const float *head, *mid, *tail;             /* [K][3] markers before smoothing */
    const uint64_t *patch_offsets, *patch_ids;  /* [K+1], [E] */
    const float *initial_normals;               /* [K][3] */
    const uint8_t *valid;                       /* [K] after the frame */
    const float *smoothed_head, *normals;       /* [K][3] */
    const float *quad_coefs, *mean_curv, *gauss_curv, *princ_curvs, *princ_dirs, *area;
    const uint32_t *nvert;
    const uint64_t *neib_ids;                   /* [E + 4K] slotted like molar_hip_membrane_state */
    const float *voro_vertexes;                 /* [E + 4K][3] */
    const float *fitted_patch_points;           /* [E][3] */
    const float *order;                         /* concatenated per tail, layout of molar_hip_lipid_tail_order */
    size_t norder;
} molar_hip_membrane_view;
/* Host destinations for molar_hip_membrane_frame_fetch: any pointer may be NULL (skipped); the E-sized arrays are
 * written up to the frame's patch_entries (see the view). */
typedef struct {
    float *head, *mid, *tail;
    uint64_t *patch_offsets, *patch_ids;
    float *initial_normals;
    uint8_t *valid;
    float *smoothed_head, *normals;
    float *quad_coefs, *mean_curv, *gauss_curv, *princ_curvs, *princ_dirs, *area;
    uint32_t *nvert;
    uint64_t *neib_ids;
    float *voro_vertexes;
    float *fitted_patch_points;
    float *order;
} molar_hip_membrane_out;
int molar_hip_membrane_plan_create(molar_hip_ctx *ctx, const molar_hip_membrane_desc *desc, molar_hip_membrane_plan **out);
void molar_hip_membrane_plan_destroy(molar_hip_membrane_plan *plan);
/* valid[nlipids] from host memory (NULL: all valid = reset_valid_lipids, lib.rs:269-273).  No frame may be in flight: with
 * a ticket pending the call changes nothing and returns MOLAR_HIP_ERR_INVALID_ARGUMENT - end the frames first (the flags
 * feed the kernels of a frame from its first launch on). */
int molar_hip_membrane_plan_set_valid(molar_hip_membrane_plan *plan, const uint8_t *valid);
/* xyz: float[natoms][3], device memory (unwrapped in place, and read until the frame ends) or host memory (uploaded;
 * the unwrapped frame is written back before _begin returns).  box9: column-major box matrix.  Returns with a ticket
 * (0 or 1) without waiting for the frame. */
int molar_hip_membrane_frame_begin(molar_hip_membrane_plan *plan, float *xyz, const float *box9, int32_t *ticket);
/* Waits for that frame; `view` may be NULL.  Errors of the frame's stages surface here (ERR_ZERO_MASS,
 * ERR_LIPID_TAIL_TOO_SHORT, ...). */
int molar_hip_membrane_frame_end(molar_hip_membrane_plan *plan, int32_t ticket, molar_hip_membrane_view *view);
/* Copies the chosen arrays of an ended frame to host memory. */
int molar_hip_membrane_frame_fetch(molar_hip_membrane_plan *plan, int32_t ticket, const molar_hip_membrane_out *out);
/* _frame_end and the fetch of per-lipid results in one wait: the chosen arrays leave on the frame's stream right behind its last
 * kernel (what LipidGroup::frame_update reads every frame, molar_membrane/src/lipid_group.rs: flags, normals, curvatures, areas,
 * vertex counts, order).  The arrays sized by the frame's patch entries (patch_ids, neib_ids, voro_vertexes,
 * fitted_patch_points) must be NULL here - their length is only known from the view; _frame_fetch brings them afterwards. */
int molar_hip_membrane_frame_end_fetch(molar_hip_membrane_plan *plan, int32_t ticket, molar_hip_membrane_view *view,
                                       const molar_hip_membrane_out *out);

/* Measure::lipid_tail_order (measure.rs:270-422), batched over `ntails` tails given as CSR:
 * tail t holds the carbons idx[tail_offsets[t] .. tail_offsets[t+1]) (n_t atoms), its normals are
 * normals[3*normal_offsets[t] .. 3*normal_offsets[t+1]) (1 or n_t-2 vectors), its n_t-1 bond orders
 * start at bond_orders[tail_offsets[t]-t], its n_t-2 results go to out[tail_offsets[t]-2t ..).
 * order_type: 0 Sz, 1 Scd, 2 ScdCorr (measure.rs:708-716).  Size violations of any tail return
 * ERR_LIPID_TAIL_TOO_SHORT / _NORMALS_COUNT / _BOND_ORDER_COUNT (measure.rs:281-291); bond order
 * counts are implied by the layout, so the third can only come from a NULL bond_orders.
 * Replaces the per-lipid rayon loop of molar_membrane (lib.rs:435-443, lipid_molecule.rs:48-59). */
int molar_hip_lipid_tail_order(molar_hip_ctx *ctx, const float *xyz, size_t natoms, const uint64_t *idx,
                               const uint64_t *tail_offsets, size_t ntails, int order_type, const float *normals,
                               const uint64_t *normal_offsets, const uint8_t *bond_orders, float *out);

/* ------------------------------------------------------------------ XTC frames (molar/src/io/xtc_handler.rs)
 *
 * Feeds the path with trajectory frames: MolAR's XtcFileHandler over the un-vendored `molly` crate
 * (read_state :64-112 -> State{coords, time, pbox}; seek_frame :200-218; seek_time :220-229 = first frame with
 * time >= t, :282-297).  GROMACS XTC, magic 1995 and 2023.  The file is indexed once at open (headers only);
 * molar_hip_xtc_read decodes `count` consecutive frames on `nthreads` host threads (0 = all cores; a frame's bit
 * stream is serial, frames are independent) into xyz[count][natoms][3] (nm), which may be host memory or device
 * memory (then through pinned staging and async copies on the context's stream; ctx may be NULL for host output).
 * box9 of a frame is column-major with columns a,b,c (Matrix3f::from_iterator(boxvec), :100).  A truncated last
 * frame ends the index (the reference reports Eof there, :325-332).  Returns ERR_IO / ERR_SIZES (frames of
 * different atom counts in one read). */
typedef struct molar_hip_xtc molar_hip_xtc;
molar_hip_xtc *molar_hip_xtc_open(const char *path);
molar_hip_xtc *molar_hip_xtc_open_memory(const void *data, size_t bytes);   /* borrowed, must outlive the handle */
void molar_hip_xtc_close(molar_hip_xtc *x);
size_t molar_hip_xtc_nframes(const molar_hip_xtc *x);
size_t molar_hip_xtc_natoms(const molar_hip_xtc *x);
int molar_hip_xtc_frame_info(const molar_hip_xtc *x, size_t frame, int32_t *natoms, int32_t *step, float *time,
                             float box9[9], float *precision);
int molar_hip_xtc_seek_time(const molar_hip_xtc *x, float t, size_t *frame);
int molar_hip_xtc_read(molar_hip_ctx *ctx, const molar_hip_xtc *x, size_t first, size_t count, float *xyz,
                       int nthreads);
/* The same window decoded ON THE DEVICE, one lane per frame (64 frames per wave): the window's compressed bytes go over the
 * link (about a third of the decoded size), `xyz_dev` (device memory, float[count][natoms][3]) is written by the kernel.
 * Bit-identical to molar_hip_xtc_read (one decoder, compiled for both sides).  A lane is far slower than a host core, so
 * this pays for windows of hundreds to thousands of frames - when the consumers of a multi-GPU node outrun the host's
 * decoder threads.  Frames whose packed triples exceed 64 bits fall back to the host decoder inside the call. */
int molar_hip_xtc_read_device(molar_hip_ctx *ctx, const molar_hip_xtc *x, size_t first, size_t count, float *xyz_dev);
/* A radial distance histogram over a block of the trajectory in one call (BASELINE config 4 for a caller without device memory
 * of its own - the process_frame loop of an RDF task, analysis_task.rs:245-252, with Histogram1D::add_one over every distance of
 * distance_search_single_pbc, molar_membrane/src/stats.rs:29-35): frames [first, first + count) are decoded on `decode_threads`
 * host threads (0 = all) into windows of 16 frames in HBM while the window before is in the fused histogram
 * (molar_hip_search_histogram_frames: selection idx - NULL = all atoms - of every frame against itself, cutoff, the frame's own
 * box from its header, periodic dimensions `pbc`); integer bins, ADDED into bins[nbins] (host) at the end.  The same sums as
 * molar_hip_xtc_read + molar_hip_search_histogram frame by frame. */
int molar_hip_xtc_histogram(molar_hip_ctx *ctx, const molar_hip_xtc *x, size_t first, size_t count, const uint64_t *idx, size_t n,
                            float cutoff, uint8_t pbc, float hmin, float hmax, size_t nbins, uint64_t *bins, int decode_threads);
/* The same between TWO selections of every frame (distance_search_double_pbc: the radial distribution of one species around
 * another); idx1 / idx2 NULL = all atoms.  Selections may overlap (an atom against itself counts at distance 0, as in the
 * reference). */
int molar_hip_xtc_histogram_double(molar_hip_ctx *ctx, const molar_hip_xtc *x, size_t first, size_t count, const uint64_t *idx1,
                                   size_t n1, const uint64_t *idx2, size_t n2, float cutoff, uint8_t pbc, float hmin, float hmax,
                                   size_t nbins, uint64_t *bins, int decode_threads);
/* Writer (xtc_handler.rs:117-168, write_state through molly::XTCWriter): ONE frame in GROMACS' compressed coordinate format
 * (magic 1995; xdrfile's algorithm: integer grid at `precision`, mixed-radix triples, runs of small deltas with an adaptive
 * delta size) into out[cap]; *out_len = bytes written, frames are simply concatenated in a file.  96 + 16 * natoms bytes
 * always suffice.  box9 as in the header (9 floats, the order the file stores).  Host memory only. */
int molar_hip_xtc_encode_frame(const float *xyz, size_t natoms, const float *box9, int32_t step, float time, float precision,
                               uint8_t *out, size_t cap, size_t *out_len);

/* ------------------------------------------------------------------ Modify (modify.rs) */

/* apply_transform :32-36 — in place on xyz (host buffers are copied back). */
int molar_hip_apply_transform(molar_hip_ctx *ctx, float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                              const float R9[9], const float t3[3]);
/* unwrap_simple_dim :40-54 */
int molar_hip_unwrap_simple(molar_hip_ctx *ctx, float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                            const float *box9, uint8_t pbc);

/* ------------------------------------------------------------------ batched per-frame analysis */

/* The per-frame align+RMSD loop of the reference's benchmark (benches/comparison_small.rs:14-25,
 * SURVEY.md §3.4) for `nframes` frames resident in one buffer frames[nframes][natoms][3]:
 *   tr = fit_transform(cur, ref); apply_transform(cur, tr); rmsd(cur, ref);
 *   center_of_mass(cur); gyration(cur)
 * Selection idx/mass are shared by all frames (topology-side arrays).  `frames` is modified in
 * place when apply != 0.  Outputs are per frame; any of them may be NULL.
 * Replaces the serial frame loop of AnalysisTask::run (analysis_task.rs:202-252) for this task. */
int molar_hip_fit_rmsd_batch(molar_hip_ctx *ctx, float *frames, size_t nframes, size_t natoms,
                             const uint64_t *idx, size_t n, const float *mass, const float *ref_xyz,
                             size_t ref_natoms, const uint64_t *ref_idx, int apply, float *rmsd_out,
                             float *R_out /*[nframes][9]*/, float *t_out /*[nframes][3]*/,
                             float *com_out /*[nframes][3]*/, float *gyr_out /*[nframes]*/);

/* The same loop for a trajectory whose frames live in HOST memory, at the rate the SELECTION crosses the link (a Rust
 * AnalysisTask hands its State's coords over frame by frame, analysis_task.rs:245-252; molar_hip_fit_rmsd_batch on a host frame
 * stages all natoms x 12 bytes from pageable memory per call).  _create packs the frame-invariant columns once (reference
 * selection, the masses of the selected atoms; all topology-side arrays in host memory) and starts `host_threads` threads
 * (0: min(8, cores / 2)) that take the selected atoms out of a frame into pinned staging.  _begin(frame) gathers, sends the packed
 * selection behind the frame before it, enqueues the batch entry's kernels on it and returns a ticket (0..2; up to three frames
 * in flight: `begin(k + 1); end(k)`); _end(ticket) waits for that frame and returns what the batch entry returns per frame
 * (each output optional) - bit-identical to it: same terms, same order.  apply != 0: the fitted selection is written into the
 * frame handed to _begin (which must stay valid, and untouched, until _end) - apply_transform (modify.rs:32-36).  Errors as for
 * the batch entry (ERR_SIZES, ERR_ZERO_MASS, ERR_SVD_FAILED from _end). */
typedef struct molar_hip_fit_stream molar_hip_fit_stream;
int molar_hip_fit_stream_create(molar_hip_ctx *ctx, size_t natoms, const uint64_t *idx, size_t n, const float *mass,
                                const float *ref_xyz, size_t ref_natoms, const uint64_t *ref_idx, int host_threads,
                                molar_hip_fit_stream **out);
int molar_hip_fit_stream_begin(molar_hip_fit_stream *stream, float *xyz, int apply, int32_t *ticket);
int molar_hip_fit_stream_end(molar_hip_fit_stream *stream, int32_t ticket, float *rmsd, float R9[9], float t3[3], float com3[3],
                             float *gyration);
void molar_hip_fit_stream_destroy(molar_hip_fit_stream *stream);

/* ------------------------------------------------------------------ batched over K selections (CSR)
 * What MolAR runs from rayon over a ParSplit (selection/system.rs:193-213, README.md:656-691): the same Measure method
 * on thousands of small sub-selections (residues, lipids, molecules).  Selection k is idx[offsets[k] .. offsets[k+1]);
 * one 64-lane wave works on each.  Outputs are per selection, caller-allocated (host or device). */

/* gyration (measure.rs:78-87); with box9 != NULL gyration_pbc (:222-232).  out[nsel]. */
int molar_hip_gyration_batch(molar_hip_ctx *ctx, const float *xyz, size_t natoms, const uint64_t *idx,
                             const uint64_t *offsets, size_t nsel, const float *mass, const float *box9, float *out);

/* rmsd (measure.rs:485-504) of selection k of frame 1 against selection k of frame 2 (idx2 NULL: same atoms); with
 * mass1 != NULL rmsd_mw (:538-558).  out[nsel]. */
int molar_hip_rmsd_batch(molar_hip_ctx *ctx, const float *xyz1, size_t natoms1, const uint64_t *idx1, const float *xyz2,
                         size_t natoms2, const uint64_t *idx2, const uint64_t *offsets, size_t nsel, const float *mass1,
                         float *out);

/* fit_transform (measure.rs:507-522) of every selection of frame 1 onto its counterpart in frame 2 (idx2 NULL: same
 * atoms, mass2 NULL: mass1); apply != 0 moves the selections of xyz1 in place (modify.rs:32-36).  Any output may be
 * NULL: R_out[nsel][9] column-major, t_out[nsel][3], and of the FITTED selection rmsd_out[nsel] (unweighted, :485-504),
 * com_out[nsel][3], gyr_out[nsel]. */
int molar_hip_fit_batch(molar_hip_ctx *ctx, float *xyz1, size_t natoms1, const uint64_t *idx1, const float *mass1,
                        const float *xyz2, size_t natoms2, const uint64_t *idx2, const float *mass2,
                        const uint64_t *offsets, size_t nsel, int apply, float *R_out, float *t_out, float *rmsd_out,
                        float *com_out, float *gyr_out);

/* ------------------------------------------------------------------ Modify::translate / rotate, principal axes */

/* translate (modify.rs:16-23): p += shift for every selected atom, in place. */
int molar_hip_translate(molar_hip_ctx *ctx, float *xyz, size_t natoms, const uint64_t *idx, size_t n, const float shift3[3]);

/* rotate (modify.rs:25-30): p <- Rotation3::from_axis_angle(unit_axis, angle) * p about the origin, in place. */
int molar_hip_rotate(molar_hip_ctx *ctx, float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                     const float unit_axis3[3], float angle);

/* principal_transform (measure.rs:102-109) / principal_transform_pbc (:246-257, box9 != NULL): the isometry
 * Translation(cm) * Rotation(axes^-1) * Translation(-cm) (:646-649) as R9 (column-major) and t3, p -> R p + t. */
int molar_hip_principal_transform(molar_hip_ctx *ctx, const float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                                  const float *mass, const float *box9, float R9[9], float t3[3]);


#ifdef __cplusplus
}
#endif
#endif /* MOLAR_HIP_H */
