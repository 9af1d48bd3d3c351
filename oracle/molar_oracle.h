/*
 * molar_oracle.h — CPU restatement of MolAR's per-frame hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under molar_amd/ (the product) may include,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / timed CPU baseline.
 *
 * PARITY PINNING: the PeriodicBox arithmetic is pinned by the reference's own
 * known answers (molar/src/periodic_box.rs:456-620, molar_python/tests/test_2.py:233-245),
 * replayed in tests/test_oracle_periodic_box.py.  Everything else on the path
 * (pair lists, rmsd, fit, gyration, inertia, lipid order) has NO asserting test or
 * golden vector in the reference that can be replayed here (tests/albumin.pdb is
 * missing from the mount; cargo/rustc are absent so the reference cannot be run):
 * for those functions this oracle is "PARITY UNPINNED" — it is a reading of the
 * Rust source, cross-checked only against an independent brute force.
 *
 * REAL is float (MolAR default, aliases.rs:10-13) or double (-DORACLE_F64, MolAR's
 * `f64` cargo feature).  Build with -ffp-contract=off: Rust never contracts a*b+c.
 *
 * Matrices are column-major REAL[9]: M(r,c) = m[c*3+r]; columns are the box vectors
 * a,b,c (periodic_box.rs:7-13).
 */
#ifndef MOLAR_ORACLE_H
#define MOLAR_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef ORACLE_F64
typedef double REAL;
#else
typedef float REAL;
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* status codes shared with include/molar_hip.h (measure.rs:732-762, periodic_box.rs:131-144) */
enum {
    ORC_OK = 0,
    ORC_ERR_SIZES = 1,
    ORC_ERR_ZERO_MASS = 2,
    ORC_ERR_SVD = 3,
    ORC_ERR_NO_PBC = 4,
    ORC_ERR_ZERO_LENGTH_VECTOR = 5,
    ORC_ERR_INVERSE_FAILED = 6,
    ORC_ERR_LIPID_TAIL_TOO_SHORT = 7,
    ORC_ERR_LIPID_NORMALS_COUNT = 8,
    ORC_ERR_LIPID_BOND_ORDER_COUNT = 9
};

#define ORC_PBC_FULL 7u
#define ORC_PBC_NONE 0u

/* periodic_box.rs:15-23 */
typedef struct {
    REAL m[9];
    REAL inv[9];
    int32_t nshift;
    REAL shifts[26 * 3];
} orc_box;

int orc_sizeof_real(void);

/* ---- periodic_box.rs ---- */
int  orc_box_from_matrix(const REAL *m9, orc_box *out);                       /* :156-176 */
int  orc_box_from_vectors_angles(REAL a, REAL b, REAL c, REAL alpha, REAL beta, REAL gamma,
                                 orc_box *out);                               /* :188-235 */
void orc_shortest_vector_dims(const orc_box *b, const REAL v[3], uint8_t dims, REAL out[3]); /* :286-318 */
REAL orc_distance_squared(const orc_box *b, const REAL p1[3], const REAL p2[3], uint8_t dims); /* :379-381 */
REAL orc_distance(const orc_box *b, const REAL p1[3], const REAL p2[3], uint8_t dims);        /* :385-387 */
void orc_closest_image_dims(const orc_box *b, const REAL p[3], const REAL target[3], uint8_t dims,
                            REAL out[3]);                                     /* :322-330 */
void orc_to_box_coords(const orc_box *b, const REAL v[3], REAL out[3]);        /* :340-344 */
void orc_to_lab_coords(const orc_box *b, const REAL v[3], REAL out[3]);        /* :356-360 */
int  orc_is_inside(const orc_box *b, const REAL p[3]);                         /* :348-352 */
void orc_box_extents(const orc_box *b, REAL out[3]);                           /* :364-366 */
void orc_lab_extents(const orc_box *b, REAL out[3]);                           /* :369-375 */
int  orc_is_triclinic(const orc_box *b);                                       /* :390-394 */
void orc_wrap_point(const orc_box *b, const REAL p[3], REAL out[3]);           /* :409-419 */

/* ---- distance_search.rs ---- */
typedef struct {
    size_t n;
    uint64_t *i;   /* DistanceSearchOutput (usize,usize,Float), distance_search.rs:6-26 */
    uint64_t *j;   /* NULL for `within` results (usize only) */
    REAL *d;       /* NULL for `within` results */
    uint64_t dims[3];   /* grid dims used (diagnostic) */
    size_t plan_len;    /* number of plan entries kept (diagnostic) */
    int threads_used;   /* threads the plan was run on: min(nthreads, entries / 3, candidate evaluations / 5e5) (diagnostic) */
} orc_pairs;

void orc_pairs_free(orc_pairs *p);

/* All searches take positions already gathered through the selection (the reference takes
 * iterators of &Pos and of ids).  pos: n*3 REAL (xyzxyz…), ids: n usize.
 * nthreads: worker threads over plan entries (mirrors rayon into_par_iter().with_min_len(3),
 * distance_search.rs:949-953); result order is plan order regardless of nthreads. */
orc_pairs *orc_search_single(REAL cutoff, const REAL *pos, const uint64_t *ids, size_t n,
                             int nthreads);                                    /* :892-915 */
orc_pairs *orc_search_single_pbc(REAL cutoff, const REAL *pos, const uint64_t *ids, size_t n,
                                 const orc_box *box, uint8_t pbc_dims, int nthreads); /* :928-954 */
orc_pairs *orc_search_double(REAL cutoff, const REAL *pos1, const uint64_t *ids1, size_t n1,
                             const REAL *pos2, const uint64_t *ids2, size_t n2,
                             int nthreads);                                    /* :659-698 */
orc_pairs *orc_search_double_pbc(REAL cutoff, const REAL *pos1, const uint64_t *ids1, size_t n1,
                                 const REAL *pos2, const uint64_t *ids2, size_t n2,
                                 const orc_box *box, uint8_t pbc_dims, int nthreads); /* :713-754 */
orc_pairs *orc_search_double_vdw(const REAL *pos1, size_t n1, const REAL *pos2, size_t n2,
                                 const REAL *vdw1, const REAL *vdw2, int nthreads);  /* :767-814 */
orc_pairs *orc_search_double_vdw_pbc(const REAL *pos1, size_t n1, const REAL *pos2, size_t n2,
                                     const REAL *vdw1, const REAL *vdw2, const orc_box *box,
                                     uint8_t pbc_dims, int nthreads);          /* :829-879 */
orc_pairs *orc_search_within(REAL cutoff, const REAL *pos1, const uint64_t *ids1, size_t n1,
                             const REAL *pos2, const uint64_t *ids2, size_t n2,
                             const REAL lower[3], const REAL upper[3], int nthreads); /* :519-558 */
orc_pairs *orc_search_within_pbc(REAL cutoff, const REAL *pos1, const uint64_t *ids1, size_t n1,
                                 const REAL *pos2, const uint64_t *ids2, size_t n2,
                                 const orc_box *box, uint8_t pbc_dims, int nthreads); /* :560-598 */

/* Independent O(N^2) checker: same predicate arithmetic, no grid.  With box==NULL plain
 * Euclidean; otherwise full minimum image (shortest_vector_dims with pbc_dims).  Emits i<j
 * in (i,j) lexicographic order of LOCAL positions, ids mapped through ids[]. */
orc_pairs *orc_brute_single(REAL cutoff, const REAL *pos, const uint64_t *ids, size_t n,
                            const orc_box *box, uint8_t pbc_dims);
orc_pairs *orc_brute_double(REAL cutoff, const REAL *pos1, const uint64_t *ids1, size_t n1,
                            const REAL *pos2, const uint64_t *ids2, size_t n2,
                            const orc_box *box, uint8_t pbc_dims);

/* bounding boxes of the non-PBC drivers (distance_search.rs:602-646) */
void orc_bounding_box_single(REAL cutoff, const REAL *pos, size_t n, REAL lower[3], REAL upper[3]);
void orc_bounding_box_double(REAL cutoff, const REAL *pos1, size_t n1, const REAL *pos2, size_t n2,
                             REAL lower[3], REAL upper[3]);

/* ---- measure.rs / modify.rs.  xyz = whole frame (AoS), idx = selection (NULL => identity),
 * n = selection length, mass = full-length column gathered through idx
 * (providers.rs:103-106,208-210). ---- */
void orc_min_max(const REAL *xyz, const uint64_t *idx, size_t n, REAL lower[3], REAL upper[3]); /* :22-36 */
void orc_center_of_geometry(const REAL *xyz, const uint64_t *idx, size_t n, REAL out[3]);       /* :39-47 */
int  orc_center_of_mass(const REAL *xyz, const uint64_t *idx, size_t n, const REAL *mass,
                        REAL out[3]);                                                           /* :60-75 */
int  orc_center_of_geometry_pbc_dims(const REAL *xyz, const uint64_t *idx, size_t n,
                                     const orc_box *b, uint8_t dims, REAL out[3]);              /* :142-168 */
int  orc_center_of_mass_pbc_dims(const REAL *xyz, const uint64_t *idx, size_t n, const REAL *mass,
                                 const orc_box *b, uint8_t dims, REAL out[3]);                  /* :172-220 */
int  orc_gyration(const REAL *xyz, const uint64_t *idx, size_t n, const REAL *mass, REAL *out); /* :78-87 */
int  orc_gyration_pbc(const REAL *xyz, const uint64_t *idx, size_t n, const REAL *mass,
                      const orc_box *b, REAL *out);                                             /* :222-232 */
int  orc_inertia(const REAL *xyz, const uint64_t *idx, size_t n, const REAL *mass,
                 REAL moments[3], REAL axes9[9]);                                               /* :90-99,573-610 */
int  orc_inertia_pbc(const REAL *xyz, const uint64_t *idx, size_t n, const REAL *mass,
                     const orc_box *b, REAL moments[3], REAL axes9[9]);                         /* :234-244 */
/* raw (pre-eigen) inertia tensor, for sign-independent comparisons */
int  orc_inertia_tensor(const REAL *xyz, const uint64_t *idx, size_t n, const REAL *mass,
                        const orc_box *b_or_null, REAL tens9[9]);
int  orc_rmsd(const REAL *xyz1, const uint64_t *idx1, size_t n1,
              const REAL *xyz2, const uint64_t *idx2, size_t n2, REAL *out);                    /* :485-504 */
int  orc_rmsd_mw(const REAL *xyz1, const uint64_t *idx1, size_t n1, const REAL *mass1,
                 const REAL *xyz2, const uint64_t *idx2, size_t n2, REAL *out);                 /* :538-558 */
/* R column-major, p -> R p + t */
int  orc_fit_transform(const REAL *xyz1, const uint64_t *idx1, size_t n1, const REAL *mass1,
                       const REAL *xyz2, const uint64_t *idx2, size_t n2, const REAL *mass2,
                       REAL R9[9], REAL t3[3]);                                                 /* :507-522 */
int  orc_fit_transform_at_origin(const REAL *xyz1, const uint64_t *idx1, size_t n1, const REAL *mass1,
                                 const REAL *xyz2, const uint64_t *idx2, size_t n2,
                                 REAL R9[9], REAL t3[3]);                                       /* :525-535 */
void orc_apply_transform(REAL *xyz, const uint64_t *idx, size_t n, const REAL R9[9],
                         const REAL t3[3]);                                             /* modify.rs:32-36 */
void orc_translate(REAL *xyz, const uint64_t *idx, size_t n, const REAL shift[3]);     /* modify.rs:16-23 */
int  orc_unwrap_simple_dim(REAL *xyz, const uint64_t *idx, size_t n, const orc_box *b,
                           uint8_t dims);                                               /* modify.rs:40-54 */
int  orc_unwrap_connectivity_dim(REAL *xyz, const uint64_t *idx, size_t n, const orc_box *b, REAL cutoff, uint8_t dims,
                                 uint64_t *group_offsets, uint64_t *group_ids, size_t *ngroups,
                                 int nthreads);                                                 /* modify.rs:72-131 */
/* order_type: 0 Sz, 1 Scd, 2 ScdCorr (measure.rs:708-716).  out has n-2 entries. */
int  orc_lipid_tail_order(const REAL *xyz, const uint64_t *idx, size_t n, int order_type,
                          const REAL *normals, size_t n_normals, const uint8_t *bond_orders,
                          size_t n_bonds, REAL *out);                                           /* :270-422 */

/* molar_membrane/src/stats.rs:13-55 Histogram1D::add_one over an array of values */
void orc_histogram_add(REAL minv, REAL maxv, size_t nbins, const REAL *vals, size_t nvals,
                       REAL *bins);

/* One iteration of molar_membrane's Membrane::smooth (lib.rs:661-812): per valid lipid a local frame from
 * its normal, quadric fit of the patch markers, Voronoi cell, curvatures, fitted normal, area; then the serial
 * scatter-average of the markers.  head [K][3], normals [K][3] and valid [K] are updated in place; lipids that
 * stay valid get coefs [K][6], mean/gauss [K], princ_curvs [K][2], princ_dirs [K][2][3], area [K], nvert [K];
 * lipid i owns slots [poff[i]+4i, poff[i+1]+4(i+1)) of neib_ids / voro ([..][3]) and uses nvert[i] of them;
 * fitted [E][3] is aligned with pids. */
int orc_membrane_smooth(const orc_box *box, size_t K, REAL *head, REAL *normals, uint8_t *valid,
                        const uint64_t *poff, const uint64_t *pids, REAL *coefs, REAL *mean_curv,
                        REAL *gauss_curv, REAL *princ_curvs, REAL *princ_dirs, REAL *area, uint32_t *nvert,
                        uint64_t *neib_ids, REAL *voro, REAL *fitted);

#ifdef __cplusplus
}
#endif
#endif
