/*
 * molar_oracle.c — CPU restatement of MolAR's hot path (see molar_oracle.h for the
 * scope, the "test infrastructure only" rule and the parity-pinning statement).
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference/molar/src/).  Arithmetic follows the Rust source operation by
 * operation in REAL precision; nothing may be contracted into an FMA
 * (compile with -ffp-contract=off).  nalgebra 0.34 (Cargo.toml:33, un-vendored)
 * supplies the small-vector kernels; their operation order as restated here:
 *   M*v      : y_r = ((M_r0*v0) + M_r1*v1) + M_r2*v2        (gemv = column axpy chain)
 *   |v|^2    : ((x*x) + (y*y)) + (z*z)                      (dot, U3 special case)
 *   inverse  : adjugate / determinant formula of linalg/inverse.rs (3x3 case)
 *   angle    : acos(clamp(a.b / (|a|*|b|), -1, 1)), 0 if either norm is 0
 */
#include "molar_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ORACLE_F64
#define R_SQRT sqrt
#define R_FLOOR floor
#define R_ABS fabs
#define R_ROUND round
#define R_TRUNC trunc
#define R_ACOS acos
#define R_COS cos
#define R_SIN sin
#define R_EPS 2.220446049250313e-16
#define R_PI 3.14159265358979323846
#define R_MAXVAL 1.7976931348623157e308
#else
#define R_SQRT sqrtf
#define R_FLOOR floorf
#define R_ABS fabsf
#define R_ROUND roundf
#define R_TRUNC truncf
#define R_ACOS acosf
#define R_COS cosf
#define R_SIN sinf
#define R_EPS 1.1920929e-07f
#define R_PI 3.14159265358979323846f
#define R_MAXVAL 3.40282347e+38f
#endif

#define M(b, r, c) ((b)[(c) * 3 + (r)])

int orc_sizeof_real(void) { return (int)sizeof(REAL); }

/* ------------------------------------------------------------------ small vectors */

static inline REAL norm2_3(const REAL v[3]) { return (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]; }
static inline REAL dot3(const REAL a[3], const REAL b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline REAL norm3(const REAL v[3]) { return R_SQRT(norm2_3(v)); }

static inline void matvec(const REAL *m, const REAL v[3], REAL out[3]) {
    REAL x = v[0], y = v[1], z = v[2];
    for (int r = 0; r < 3; ++r) out[r] = (M(m, r, 0) * x + M(m, r, 1) * y) + M(m, r, 2) * z;
}

static inline void cross3(const REAL a[3], const REAL b[3], REAL o[3]) {
    REAL x = a[1] * b[2] - a[2] * b[1];
    REAL y = a[2] * b[0] - a[0] * b[2];
    REAL z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}

static inline void normalize3(const REAL a[3], REAL o[3]) {
    REAL n = norm3(a);
    o[0] = a[0] / n; o[1] = a[1] / n; o[2] = a[2] / n;
}

static inline REAL angle3(const REAL a[3], const REAL b[3]) {
    REAL prod = dot3(a, b);
    REAL n1 = norm3(a), n2 = norm3(b);
    if (n1 == 0 || n2 == 0) return 0;
    REAL c = prod / (n1 * n2);
    if (c < (REAL)-1) c = (REAL)-1;
    if (c > (REAL)1) c = (REAL)1;
    return R_ACOS(c);
}

/* Rust `x as usize`: saturating, NaN -> 0 */
static inline uint64_t as_usize(REAL x) {
    if (!(x > 0)) return 0;
    if (x >= (REAL)18446744073709551615.0) return UINT64_MAX;
    return (uint64_t)x;
}
/* Rust `x as isize` */
static inline int64_t as_isize(REAL x) {
    if (x != x) return 0;
    if (x >= (REAL)9223372036854775807.0) return INT64_MAX;
    if (x <= (REAL)-9223372036854775808.0) return INT64_MIN;
    return (int64_t)x;
}

/* ------------------------------------------------------------------ periodic_box.rs */

/* periodic_box.rs:25-66 */
static void build_tric_corrections(orc_box *bx) {
    const REAL *m = bx->m;
    bx->nshift = 0;
    if (M(m, 0, 1) == 0 && M(m, 0, 2) == 0 && M(m, 1, 0) == 0 && M(m, 1, 2) == 0 && M(m, 2, 0) == 0 &&
        M(m, 2, 1) == 0)
        return;
    REAL a[3], b[3], c[3], t[3];
    for (int r = 0; r < 3; ++r) { a[r] = M(m, r, 0); b[r] = M(m, r, 1); c[r] = M(m, r, 2); }
    REAL n0, n1, n2, n3;
    for (int r = 0; r < 3; ++r) t[r] = (a[r] + b[r]) + c[r];
    n0 = norm3(t);
    for (int r = 0; r < 3; ++r) t[r] = (a[r] + b[r]) - c[r];
    n1 = norm3(t);
    for (int r = 0; r < 3; ++r) t[r] = (a[r] - b[r]) + c[r];
    n2 = norm3(t);
    for (int r = 0; r < 3; ++r) t[r] = (-a[r] + b[r]) + c[r];
    n3 = norm3(t);
    REAL mx = n0;               /* f32::max chain (:46-49) */
    if (n1 > mx) mx = n1;
    if (n2 > mx) mx = n2;
    if (n3 > mx) mx = n3;
    REAL half_diag = (REAL)0.5 * mx;
    REAL two_hd = (REAL)2.0 * half_diag;
    REAL bound2 = two_hd * two_hd;                      /* powi(2) */
    for (int i = -1; i <= 1; ++i)
        for (int j = -1; j <= 1; ++j)
            for (int k = -1; k <= 1; ++k) {
                if (i == 0 && j == 0 && k == 0) continue;
                REAL s[3];
                for (int r = 0; r < 3; ++r) s[r] = ((REAL)i * a[r] + (REAL)j * b[r]) + (REAL)k * c[r];
                if (norm2_3(s) < bound2) {
                    memcpy(&bx->shifts[3 * bx->nshift], s, sizeof s);
                    bx->nshift++;
                }
            }
}

/* nalgebra linalg/inverse.rs, 3x3 branch (called from periodic_box.rs:167-169) */
static int inverse3(const REAL *m, REAL *o) {
    REAL m11 = M(m, 0, 0), m12 = M(m, 0, 1), m13 = M(m, 0, 2);
    REAL m21 = M(m, 1, 0), m22 = M(m, 1, 1), m23 = M(m, 1, 2);
    REAL m31 = M(m, 2, 0), m32 = M(m, 2, 1), m33 = M(m, 2, 2);
    REAL minor_m12_m23 = m22 * m33 - m32 * m23;
    REAL minor_m11_m23 = m21 * m33 - m31 * m23;
    REAL minor_m11_m22 = m21 * m32 - m31 * m22;
    REAL det = (m11 * minor_m12_m23 - m12 * minor_m11_m23) + m13 * minor_m11_m22;
    if (det == 0) return 0;
    M(o, 0, 0) = minor_m12_m23 / det;
    M(o, 0, 1) = (m13 * m32 - m33 * m12) / det;
    M(o, 0, 2) = (m12 * m23 - m22 * m13) / det;
    M(o, 1, 0) = -minor_m11_m23 / det;
    M(o, 1, 1) = (m11 * m33 - m31 * m13) / det;
    M(o, 1, 2) = (m13 * m21 - m23 * m11) / det;
    M(o, 2, 0) = minor_m11_m22 / det;
    M(o, 2, 1) = (m12 * m31 - m32 * m11) / det;
    M(o, 2, 2) = (m11 * m22 - m21 * m12) / det;
    return 1;
}

/* periodic_box.rs:156-176 */
int orc_box_from_matrix(const REAL *m9, orc_box *out) {
    for (int c = 0; c < 3; ++c) {
        REAL col[3] = {M(m9, 0, c), M(m9, 1, c), M(m9, 2, c)};
        if (norm3(col) == 0) return ORC_ERR_ZERO_LENGTH_VECTOR;
    }
    memcpy(out->m, m9, 9 * sizeof(REAL));
    if (!inverse3(out->m, out->inv)) return ORC_ERR_INVERSE_FAILED;
    build_tric_corrections(out);
    return ORC_OK;
}

/* periodic_box.rs:188-235 (AngleTooSmall reported as INVERSE_FAILED+100 is avoided: use 10) */
int orc_box_from_vectors_angles(REAL a, REAL b, REAL c, REAL alpha, REAL beta, REAL gamma, orc_box *out) {
    REAL m[9] = {0};
    if (a == 0 || b == 0 || c == 0) return ORC_ERR_ZERO_LENGTH_VECTOR;
    if (alpha < 60 || beta < 60 || gamma < 60) return 10;
    M(m, 0, 0) = a;
    if (alpha != 90 || beta != 90 || gamma != 90) {
        const REAL d2r = R_PI / (REAL)180.0;
        REAL cosa = alpha != 90 ? R_COS(alpha * d2r) : 0;
        REAL cosb = beta != 90 ? R_COS(beta * d2r) : 0;
        REAL sing = 1, cosg = 0;
        if (gamma != 90) { sing = R_SIN(gamma * d2r); cosg = R_COS(gamma * d2r); }
        M(m, 0, 1) = b * cosg;
        M(m, 1, 1) = b * sing;
        M(m, 0, 2) = c * cosb;
        M(m, 1, 2) = c * (cosa - cosb * cosg) / sing;
        M(m, 2, 2) = R_SQRT(c * c - M(m, 0, 2) * M(m, 0, 2) - M(m, 1, 2) * M(m, 1, 2));
    } else {
        M(m, 1, 1) = b;
        M(m, 2, 2) = c;
    }
    return orc_box_from_matrix(m, out);
}

/* periodic_box.rs:286-318 */
void orc_shortest_vector_dims(const orc_box *b, const REAL v[3], uint8_t dims, REAL out[3]) {
    REAL f[3], start[3];
    matvec(b->inv, v, f);
    for (int i = 0; i < 3; ++i)
        if (dims & (1u << i)) f[i] -= R_ROUND(f[i]);
    matvec(b->m, f, start);
    if (b->nshift == 0 || dims != ORC_PBC_FULL) {
        out[0] = start[0]; out[1] = start[1]; out[2] = start[2];
        return;
    }
    REAL best[3] = {start[0], start[1], start[2]};
    REAL best2 = norm2_3(start);
    for (int k = 0; k < b->nshift; ++k) {
        REAL cand[3] = {start[0] + b->shifts[3 * k], start[1] + b->shifts[3 * k + 1], start[2] + b->shifts[3 * k + 2]};
        REAL n2 = norm2_3(cand);
        if (n2 < best2) { best2 = n2; best[0] = cand[0]; best[1] = cand[1]; best[2] = cand[2]; }
    }
    out[0] = best[0]; out[1] = best[1]; out[2] = best[2];
}

/* periodic_box.rs:379-381 */
REAL orc_distance_squared(const orc_box *b, const REAL p1[3], const REAL p2[3], uint8_t dims) {
    REAL v[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, s[3];
    orc_shortest_vector_dims(b, v, dims, s);
    return norm2_3(s);
}

/* periodic_box.rs:385-387 */
REAL orc_distance(const orc_box *b, const REAL p1[3], const REAL p2[3], uint8_t dims) {
    return R_SQRT(orc_distance_squared(b, p1, p2, dims));
}

/* periodic_box.rs:322-330 */
void orc_closest_image_dims(const orc_box *b, const REAL p[3], const REAL target[3], uint8_t dims, REAL out[3]) {
    REAL v[3] = {p[0] - target[0], p[1] - target[1], p[2] - target[2]}, s[3];
    orc_shortest_vector_dims(b, v, dims, s);
    out[0] = target[0] + s[0]; out[1] = target[1] + s[1]; out[2] = target[2] + s[2];
}

void orc_to_box_coords(const orc_box *b, const REAL v[3], REAL out[3]) { matvec(b->inv, v, out); }
void orc_to_lab_coords(const orc_box *b, const REAL v[3], REAL out[3]) { matvec(b->m, v, out); }

/* periodic_box.rs:348-352 */
int orc_is_inside(const orc_box *b, const REAL p[3]) {
    REAL v[3];
    matvec(b->inv, p, v);
    return v[0] < 1 && v[1] < 1 && v[2] < 1 && v[0] >= 0 && v[1] >= 0 && v[2] >= 0;
}

/* periodic_box.rs:364-366 */
void orc_box_extents(const orc_box *b, REAL out[3]) {
    for (int c = 0; c < 3; ++c) {
        REAL col[3] = {M(b->m, 0, c), M(b->m, 1, c), M(b->m, 2, c)};
        out[c] = norm3(col);
    }
}

/* periodic_box.rs:369-375 — ROW sums of the matrix */
void orc_lab_extents(const orc_box *b, REAL out[3]) {
    for (int r = 0; r < 3; ++r) out[r] = (M(b->m, r, 0) + M(b->m, r, 1)) + M(b->m, r, 2);
}

int orc_is_triclinic(const orc_box *b) {
    const REAL *m = b->m;
    return M(m, 0, 1) != 0 || M(m, 0, 2) != 0 || M(m, 1, 0) != 0 || M(m, 1, 2) != 0 || M(m, 2, 0) != 0 || M(m, 2, 1) != 0;
}

/* periodic_box.rs:409-419 (note the reference's `1.0 - bv` for negatives) */
void orc_wrap_point(const orc_box *b, const REAL p[3], REAL out[3]) {
    REAL bv[3];
    matvec(b->inv, p, bv);
    for (int i = 0; i < 3; ++i) {
        bv[i] = bv[i] - R_TRUNC(bv[i]);           /* f32::fract */
        if (bv[i] < 0) bv[i] = (REAL)1.0 - bv[i];
    }
    matvec(b->m, bv, out);
}

/* ------------------------------------------------------------------ grid (distance_search.rs:33-215) */

typedef struct {
    uint64_t id;
    REAL p[3];
} gitem;

typedef struct {
    uint64_t dims[3];
    size_t ncells;
    size_t *start;   /* ncells+1 */
    gitem *items;    /* in-cell order = reference push order */
    size_t nitems;
} grid_t;

static void grid_free(grid_t *g) {
    free(g->start);
    free(g->items);
    g->start = NULL;
    g->items = NULL;
}

/* distance_search.rs:103-110 */
static int grid_dims_from_extents(REAL cutoff, const REAL ext[3], uint64_t dims[3]) {
    for (int d = 0; d < 3; ++d) {
        uint64_t s = as_usize(R_FLOOR(ext[d] / cutoff));
        if (s < 1) s = 1;
        dims[d] = s;
    }
    /* refuse absurd grids instead of exhausting memory (the reference would try to allocate) */
    double tot = (double)dims[0] * (double)dims[1] * (double)dims[2];
    return tot <= 2.0e9;
}

static inline size_t loc_to_ind(const uint64_t dims[3], const uint64_t loc[3]) {   /* :85-87 */
    return loc[0] + loc[1] * dims[0] + loc[2] * dims[0] * dims[1];
}

/* Build CSR from (cell, item) lists.  `order` phase 0 items come first in each cell, then phase 1
 * (wrapped atoms), each in input order — the reference's push order (:180, :203-209). */
static void grid_build(grid_t *g, size_t n, const size_t *cell, const uint8_t *phase, const gitem *src) {
    g->ncells = (size_t)(g->dims[0] * g->dims[1] * g->dims[2]);
    g->start = (size_t *)calloc(g->ncells + 1, sizeof(size_t));
    size_t kept = 0;
    for (size_t k = 0; k < n; ++k)
        if (cell[k] != (size_t)-1) { g->start[cell[k] + 1]++; kept++; }
    for (size_t c = 0; c < g->ncells; ++c) g->start[c + 1] += g->start[c];
    g->items = (gitem *)malloc((kept ? kept : 1) * sizeof(gitem));
    g->nitems = kept;
    size_t *cur = (size_t *)malloc((g->ncells ? g->ncells : 1) * sizeof(size_t));
    memcpy(cur, g->start, g->ncells * sizeof(size_t));
    for (int ph = 0; ph < 2; ++ph)
        for (size_t k = 0; k < n; ++k)
            if (cell[k] != (size_t)-1 && phase[k] == ph) g->items[cur[cell[k]]++] = src[k];
    free(cur);
}

/* distance_search.rs:120-142 */
static void grid_populate(grid_t *g, const REAL *pos, const uint64_t *ids, size_t n, const REAL lower[3],
                          const REAL upper[3]) {
    REAL dim_sz[3] = {upper[0] - lower[0], upper[1] - lower[1], upper[2] - lower[2]};
    size_t *cell = (size_t *)malloc((n ? n : 1) * sizeof(size_t));
    uint8_t *phase = (uint8_t *)calloc(n ? n : 1, 1);
    gitem *src = (gitem *)malloc((n ? n : 1) * sizeof(gitem));
    for (size_t k = 0; k < n; ++k) {
        const REAL *p = pos + 3 * k;
        uint64_t loc[3];
        int ok = 1;
        for (int d = 0; d < 3; ++d) {
            int64_t nn = as_isize(R_FLOOR((REAL)g->dims[d] * (p[d] - lower[d]) / dim_sz[d]));
            if (nn < 0 || nn >= (int64_t)g->dims[d]) { ok = 0; break; }
            loc[d] = (uint64_t)nn;
        }
        cell[k] = ok ? loc_to_ind(g->dims, loc) : (size_t)-1;
        src[k].id = ids ? ids[k] : k;
        src[k].p[0] = p[0]; src[k].p[1] = p[1]; src[k].p[2] = p[2];
    }
    grid_build(g, n, cell, phase, src);
    free(cell); free(phase); free(src);
}

/* distance_search.rs:144-210 */
static void grid_populate_pbc(grid_t *g, const REAL *pos, const uint64_t *ids, size_t n, const orc_box *box,
                              uint8_t pbc) {
    size_t *cell = (size_t *)malloc((n ? n : 1) * sizeof(size_t));
    uint8_t *phase = (uint8_t *)calloc(n ? n : 1, 1);
    gitem *src = (gitem *)malloc((n ? n : 1) * sizeof(gitem));
    for (size_t k = 0; k < n; ++k) {
        const REAL *p = pos + 3 * k;
        REAL rel[3];
        matvec(box->inv, p, rel);                                  /* :156 */
        uint64_t loc[3];
        int correct = 1, drop = 0;
        for (int d = 0; d < 3; ++d) {                              /* :161-171 */
            if (rel[d] < 0 || rel[d] >= 1) {
                if (!(pbc & (1u << d))) { drop = 1; break; }
                correct = 0;
                break;
            }
        }
        src[k].id = ids ? ids[k] : k;
        if (drop) { cell[k] = (size_t)-1; continue; }
        if (correct) {                                             /* :173-180 */
            for (int d = 0; d < 3; ++d) {
                uint64_t l = as_usize(R_FLOOR(rel[d] * (REAL)g->dims[d]));
                if (l > g->dims[d] - 1) l = g->dims[d] - 1;
                loc[d] = l;
            }
            cell[k] = loc_to_ind(g->dims, loc);
            src[k].p[0] = p[0]; src[k].p[1] = p[1]; src[k].p[2] = p[2];
        } else {                                                   /* :181-199 */
            for (int d = 0; d < 3; ++d) {
                if (pbc & (1u << d)) {
                    rel[d] = rel[d] - R_TRUNC(rel[d]);             /* fract */
                    if (rel[d] < 0) rel[d] = (REAL)1.0 + rel[d];
                }
                uint64_t l = as_usize(R_FLOOR(rel[d] * (REAL)g->dims[d]));
                if (l > g->dims[d] - 1) l = g->dims[d] - 1;
                loc[d] = l;
            }
            matvec(box->m, rel, src[k].p);                          /* :196 */
            cell[k] = loc_to_ind(g->dims, loc);
            phase[k] = 1;                                          /* appended after in-box atoms :203-209 */
        }
    }
    grid_build(g, n, cell, phase, src);
    free(cell); free(phase); free(src);
}

/* ------------------------------------------------------------------ plan (distance_search.rs:39-60,217-269) */

static const uint8_t MASK[14][2][3] = {
    {{0, 0, 0}, {0, 0, 0}},
    {{0, 0, 0}, {1, 0, 0}}, {{0, 0, 0}, {0, 1, 0}}, {{0, 0, 0}, {0, 0, 1}},
    {{0, 0, 0}, {1, 1, 0}}, {{0, 0, 0}, {1, 0, 1}}, {{0, 0, 0}, {0, 1, 1}},
    {{0, 0, 0}, {1, 1, 1}},
    {{1, 0, 0}, {0, 1, 0}}, {{1, 0, 0}, {0, 0, 1}}, {{0, 1, 0}, {0, 0, 1}},
    {{1, 1, 0}, {0, 0, 1}}, {{1, 0, 1}, {0, 1, 0}}, {{0, 1, 1}, {1, 0, 0}},
};

typedef struct {
    size_t c1, c2;
    uint8_t wrap;
} plan_item;

static inline size_t cell_len(const grid_t *g, size_t c) { return g->start[c + 1] - g->start[c]; }

static plan_item *search_plan(const grid_t *g1, const grid_t *g2, uint8_t pbc, size_t *plan_len) {
    plan_item *plan = (plan_item *)malloc((14 * g1->ncells + 1) * sizeof(plan_item));
    size_t np = 0;
    for (uint64_t x = 0; x < g1->dims[0]; ++x)
        for (uint64_t y = 0; y < g1->dims[1]; ++y)
            for (uint64_t z = 0; z < g1->dims[2]; ++z)
                for (int m = 0; m < 14; ++m) {
                    uint64_t c[2][3] = {{x + MASK[m][0][0], y + MASK[m][0][1], z + MASK[m][0][2]},
                                        {x + MASK[m][1][0], y + MASK[m][1][1], z + MASK[m][1][2]}};
                    uint8_t wrapped = 0;
                    int skip = 0;
                    for (int i = 0; i < 2 && !skip; ++i)
                        for (int d = 0; d < 3; ++d)
                            if (c[i][d] == g1->dims[d]) {
                                if (pbc & (1u << d)) { c[i][d] = 0; wrapped |= (uint8_t)(1u << d); }
                                else { skip = 1; break; }
                            }
                    if (skip) continue;
                    size_t i1 = loc_to_ind(g1->dims, c[0]), i2 = loc_to_ind(g1->dims, c[1]);
                    int keep;
                    if (g2)
                        keep = (cell_len(g1, i1) > 0 && cell_len(g2, i2) > 0) ||
                               (cell_len(g2, i1) > 0 && cell_len(g1, i2) > 0);
                    else
                        keep = cell_len(g1, i1) > 0 && cell_len(g1, i2) > 0;
                    if (keep) { plan[np].c1 = i1; plan[np].c2 = i2; plan[np].wrap = wrapped; np++; }
                }
    *plan_len = np;
    return plan;
}

/* ------------------------------------------------------------------ growable output */

typedef struct {
    size_t n, cap;
    size_t cap_jd;         /* entries the j / d planes hold: a `within` search grows the i plane alone */
    uint64_t *i, *j;
    REAL *d;
    int with_jd;
} pvec;

static inline void pv_push(pvec *v, uint64_t i, uint64_t j, REAL d) {
    if (v->n == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 64;
        v->i = (uint64_t *)realloc(v->i, nc * sizeof(uint64_t));
        if (v->with_jd) {
            v->j = (uint64_t *)realloc(v->j, nc * sizeof(uint64_t));
            v->d = (REAL *)realloc(v->d, nc * sizeof(REAL));
            v->cap_jd = nc;
        }
        v->cap = nc;
    }
    v->i[v->n] = i;
    if (v->with_jd) { v->j[v->n] = j; v->d[v->n] = d; }
    v->n++;
}

/* ------------------------------------------------------------------ cell-pair kernels (distance_search.rs:271-517) */

enum { K_SINGLE, K_DOUBLE, K_WITHIN, K_VDW };

typedef struct {
    int kind;
    int use_pbc;           /* the *_pbc variants: wrapped pairs use box->distance_squared */
    REAL cutoff2;
    const grid_t *g1, *g2;
    const orc_box *box;
    const REAL *vdw1, *vdw2;
} sctx;

static inline REAL pair_d2(const sctx *s, const REAL *p1, const REAL *p2, uint8_t wrap) {
    if (s->use_pbc && wrap) return orc_distance_squared(s->box, p1, p2, wrap);   /* :485-486 */
    REAL v[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};                    /* :488 */
    return norm2_3(v);
}

/* search_cell_pair_single(_pbc) :432-517 */
static void cell_pair_single(const sctx *s, plan_item pr, pvec *found) {
    const grid_t *g = s->g1;
    const gitem *a = g->items + g->start[pr.c1];
    size_t n1 = cell_len(g, pr.c1);
    if (pr.c1 == pr.c2) {
        for (size_t i = 0; i + 1 < n1; ++i)
            for (size_t j = i + 1; j < n1; ++j) {
                REAL d2 = pair_d2(s, a[i].p, a[j].p, pr.wrap);
                if (d2 <= s->cutoff2) pv_push(found, a[i].id, a[j].id, R_SQRT(d2));
            }
    } else {
        const gitem *b = g->items + g->start[pr.c2];
        size_t n2 = cell_len(g, pr.c2);
        for (size_t i = 0; i < n1; ++i)
            for (size_t j = 0; j < n2; ++j) {
                REAL d2 = pair_d2(s, a[i].p, b[j].p, pr.wrap);
                if (d2 <= s->cutoff2) pv_push(found, a[i].id, b[j].id, R_SQRT(d2));
            }
    }
}

/* search_cell_pair_double(_pbc) :324-373, _vdw(_pbc) :375-430, _within(_pbc) :271-322.
 * cA indexes grid1, cB indexes grid2. */
static void cell_pair_two(const sctx *s, size_t cA, size_t cB, uint8_t wrap, pvec *found) {
    const gitem *a = s->g1->items + s->g1->start[cA];
    const gitem *b = s->g2->items + s->g2->start[cB];
    size_t n1 = cell_len(s->g1, cA), n2 = cell_len(s->g2, cB);
    for (size_t i = 0; i < n1; ++i)
        for (size_t j = 0; j < n2; ++j) {
            REAL d2 = pair_d2(s, a[i].p, b[j].p, wrap);
            if (s->kind == K_VDW) {
                REAL cutoff = (s->vdw1[a[i].id] + s->vdw2[b[j].id]) + (REAL)R_EPS;   /* :392,:423 */
                if (d2 <= cutoff * cutoff) pv_push(found, a[i].id, b[j].id, R_SQRT(d2));
            } else if (d2 <= s->cutoff2) {
                if (s->kind == K_WITHIN) { pv_push(found, a[i].id, 0, 0); break; }    /* :287-290 */
                pv_push(found, a[i].id, b[j].id, R_SQRT(d2));
            }
        }
}

/* Allocator caches.  The reference's per-task Vecs and its collected result come out of an allocator that keeps what it
 * has handed out before (rayon workers' thread-local caches; a freed 4 GB block stays mapped): a second frame of a trajectory
 * does not fault its 7 GB of pages in again.  Plain malloc / free of blocks this size returns them to the kernel every time,
 * and with 256 threads faulting pages in at once the baseline measured the page-fault path, not the search.  So the arenas
 * (one per thread) and ONE set of result buffers are kept between calls and grow only; orc_pairs_free hands the result
 * buffers back to the cache.  Not thread-safe across concurrent searches (the tests and the benchmark run one at a time). */
static pvec *g_arena = NULL;
static int g_arena_n = 0;
static pvec *arena_pool(int nt) {
    if (nt > g_arena_n) {
        g_arena = (pvec *)realloc(g_arena, (size_t)nt * sizeof(pvec));
        memset(g_arena + g_arena_n, 0, (size_t)(nt - g_arena_n) * sizeof(pvec));
        g_arena_n = nt;
    }
    return g_arena;
}
static void arena_want_jd(pvec *v) {          /* an arena a `within` search has used (and grown) has no, or too short, j / d planes */
    if (v->with_jd && v->cap_jd < v->cap) {
        v->j = (uint64_t *)realloc(v->j, v->cap * sizeof(uint64_t));
        v->d = (REAL *)realloc(v->d, v->cap * sizeof(REAL));
        v->cap_jd = v->cap;
    }
}
static struct { uint64_t *i, *j; REAL *d; size_t cap_i, cap_jd; int busy; } g_res;
static void result_buffers(orc_pairs *out, size_t tot, int with_jd) {
    const size_t want = tot ? tot : 1;
    if (g_res.busy) {                       /* a result is still out: plain allocation for this one */
        out->i = (uint64_t *)malloc(want * sizeof(uint64_t));
        if (with_jd) { out->j = (uint64_t *)malloc(want * sizeof(uint64_t)); out->d = (REAL *)malloc(want * sizeof(REAL)); }
        return;
    }
    if (g_res.cap_i < want) { free(g_res.i); g_res.i = (uint64_t *)malloc(want * sizeof(uint64_t)); g_res.cap_i = want; }
    if (with_jd && g_res.cap_jd < want) {
        free(g_res.j); free(g_res.d);
        g_res.j = (uint64_t *)malloc(want * sizeof(uint64_t));
        g_res.d = (REAL *)malloc(want * sizeof(REAL));
        g_res.cap_jd = want;
    }
    out->i = g_res.i;
    if (with_jd) { out->j = g_res.j; out->d = g_res.d; }
    g_res.busy = 1;
}

/* driver tail: plan.into_par_iter().with_min_len(3).map(..).flatten().collect()
 * (distance_search.rs:542-557, 949-953): ordered concatenation of per-entry results.
 *
 * Schedule (what bench.py's cpu_baseline and the tools' CPU columns time):
 *  - rayon never splits below 3 plan entries and adapts its splitting to the work it finds; the restatement uses
 *    min(nthreads, entries / 3, candidate evaluations / 5e5) threads and stays serial for one - a plan of a few hundred
 *    small entries (a 20-atom selection in a 100k-atom box) costs 8 ms serially, and forking a team of 256 threads over
 *    it used to cost 250 ms whatever the problem (round 4's CPU columns);
 *  - every thread appends the results of its entries to ONE growing arena of its own (the reference's Vec per task draws
 *    on rayon's thread-local allocator caches; a malloc + realloc chain per plan entry made the 1M-atom baseline
 *    page-fault-bound) and records (thread, start, length) per entry; the ordered concatenation is a prefix sum over the
 *    entries and one parallel copy. */
typedef struct { int th; size_t start, n; } part_ref;

static size_t plan_work(const sctx *s, const plan_item *plan, size_t np) {
    size_t w = 0;
    for (size_t e = 0; e < np; ++e) {
        if (s->kind == K_SINGLE) {
            w += cell_len(s->g1, plan[e].c1) * cell_len(s->g1, plan[e].c2);
        } else {
            w += cell_len(s->g1, plan[e].c1) * cell_len(s->g2, plan[e].c2) + cell_len(s->g1, plan[e].c2) * cell_len(s->g2, plan[e].c1);
        }
    }
    return w;
}

static orc_pairs *run_plan(const sctx *s, const plan_item *plan, size_t np, int nthreads) {
    int with_jd = s->kind != K_WITHIN;
    int nt = nthreads > 0 ? nthreads : 1;
#ifdef _OPENMP
    {
        size_t by_len = np / 3, by_work = plan_work(s, plan, np) / 500000u;
        if ((size_t)nt > by_len) nt = (int)by_len;
        if ((size_t)nt > by_work) nt = (int)by_work;
        if (nt < 1) nt = 1;
    }
#else
    nt = 1;
#endif
    pvec *arena = arena_pool(nt);
    part_ref *ref = (part_ref *)calloc(np ? np : 1, sizeof(part_ref));
    for (int t = 0; t < nt; ++t) { arena[t].with_jd = with_jd; arena[t].n = 0; arena_want_jd(&arena[t]); }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 3) num_threads(nt) if (nt > 1)
#endif
    for (long e = 0; e < (long)np; ++e) {
#ifdef _OPENMP
        const int th = omp_get_thread_num();
#else
        const int th = 0;
#endif
        pvec *f = &arena[th];
        ref[e].th = th;
        ref[e].start = f->n;
        if (s->kind == K_SINGLE) {
            cell_pair_single(s, plan[e], f);
        } else {
            cell_pair_two(s, plan[e].c1, plan[e].c2, plan[e].wrap, f);
            cell_pair_two(s, plan[e].c2, plan[e].c1, plan[e].wrap, f);   /* (pair.1, pair.0, pair.2) */
        }
        ref[e].n = f->n - ref[e].start;
    }
    orc_pairs *out = (orc_pairs *)calloc(1, sizeof(orc_pairs));
    size_t *off = (size_t *)malloc((np + 1) * sizeof(size_t));
    off[0] = 0;
    for (size_t e = 0; e < np; ++e) off[e + 1] = off[e] + ref[e].n;
    size_t tot = off[np];
    out->n = tot;
    result_buffers(out, tot, with_jd);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt) if (nt > 1)
#endif
    for (long e = 0; e < (long)np; ++e) {
        const pvec *f = &arena[ref[e].th];
        const size_t st = ref[e].start, n = ref[e].n;
        if (n) {
            memcpy(out->i + off[e], f->i + st, n * sizeof(uint64_t));
            if (with_jd) {
                memcpy(out->j + off[e], f->j + st, n * sizeof(uint64_t));
                memcpy(out->d + off[e], f->d + st, n * sizeof(REAL));
            }
        }
    }
    free(ref);
    free(off);
    out->plan_len = np;
    out->threads_used = nt;
    return out;
}

void orc_pairs_free(orc_pairs *p) {
    if (!p) return;
    if (p->i == g_res.i && p->i) {          /* the cached buffers go back to the cache */
        g_res.busy = 0;
    } else {
        free(p->i); free(p->j); free(p->d);
    }
    free(p);
}

static orc_pairs *empty_result(void) {
    orc_pairs *out = (orc_pairs *)calloc(1, sizeof(orc_pairs));
    out->i = (uint64_t *)malloc(sizeof(uint64_t));
    return out;
}

/* distance_search.rs:602-616: seeded with ZEROS, so the box always contains the origin */
static void compute_min_max0(const REAL *pos, size_t n, REAL lower[3], REAL upper[3]) {
    for (int d = 0; d < 3; ++d) lower[d] = upper[d] = 0;
    for (size_t k = 0; k < n; ++k)
        for (int d = 0; d < 3; ++d) {
            REAL v = pos[3 * k + d];
            if (v < lower[d]) lower[d] = v;
            if (v > upper[d]) upper[d] = v;
        }
}

/* distance_search.rs:638-646 */
void orc_bounding_box_single(REAL cutoff, const REAL *pos, size_t n, REAL lower[3], REAL upper[3]) {
    compute_min_max0(pos, n, lower, upper);
    for (int d = 0; d < 3; ++d) {
        lower[d] += (-cutoff - (REAL)R_EPS);
        upper[d] += (cutoff + (REAL)R_EPS);
    }
}

/* distance_search.rs:618-636 */
void orc_bounding_box_double(REAL cutoff, const REAL *pos1, size_t n1, const REAL *pos2, size_t n2, REAL lower[3],
                             REAL upper[3]) {
    REAL l1[3], u1[3], l2[3], u2[3];
    compute_min_max0(pos1, n1, l1, u1);
    compute_min_max0(pos2, n2, l2, u2);
    for (int d = 0; d < 3; ++d) {
        lower[d] = l1[d] < l2[d] ? l1[d] : l2[d];    /* f32::min */
        upper[d] = u1[d] > u2[d] ? u1[d] : u2[d];
        lower[d] += (-cutoff - (REAL)R_EPS);
        upper[d] += (cutoff + (REAL)R_EPS);
    }
}

static orc_pairs *finish(orc_pairs *r, const grid_t *g) {
    r->dims[0] = g->dims[0]; r->dims[1] = g->dims[1]; r->dims[2] = g->dims[2];
    return r;
}

static orc_pairs *search_two(int kind, REAL cutoff, const REAL *pos1, const uint64_t *ids1, size_t n1,
                             const REAL *pos2, const uint64_t *ids2, size_t n2, const REAL *lower,
                             const REAL *upper, const orc_box *box, uint8_t pbc, const REAL *vdw1,
                             const REAL *vdw2, int nthreads) {
    grid_t g1 = {0}, g2 = {0};
    int ok;
    if (box) {
        REAL ext[3];
        orc_lab_extents(box, ext);
        ok = grid_dims_from_extents(cutoff, ext, g1.dims);            /* :116-118 */
    } else {
        REAL ext[3] = {upper[0] - lower[0], upper[1] - lower[1], upper[2] - lower[2]};
        ok = grid_dims_from_extents(cutoff, ext, g1.dims);            /* :112-114 */
    }
    if (!ok) return NULL;
    memcpy(g2.dims, g1.dims, sizeof g1.dims);
    if (box) {
        grid_populate_pbc(&g1, pos1, ids1, n1, box, pbc);
        grid_populate_pbc(&g2, pos2, ids2, n2, box, pbc);
    } else {
        grid_populate(&g1, pos1, ids1, n1, lower, upper);
        grid_populate(&g2, pos2, ids2, n2, lower, upper);
    }
    size_t np;
    plan_item *plan = search_plan(&g1, &g2, box ? pbc : ORC_PBC_NONE, &np);
    sctx s = {kind, box != NULL, cutoff * cutoff, &g1, &g2, box, vdw1, vdw2};
    orc_pairs *r = finish(run_plan(&s, plan, np, nthreads), &g1);
    free(plan);
    grid_free(&g1);
    grid_free(&g2);
    return r;
}

/* distance_search.rs:892-915 */
orc_pairs *orc_search_single(REAL cutoff, const REAL *pos, const uint64_t *ids, size_t n, int nthreads) {
    REAL lower[3], upper[3];
    orc_bounding_box_single(cutoff, pos, n, lower, upper);
    grid_t g = {0};
    REAL ext[3] = {upper[0] - lower[0], upper[1] - lower[1], upper[2] - lower[2]};
    if (!grid_dims_from_extents(cutoff, ext, g.dims)) return NULL;
    grid_populate(&g, pos, ids, n, lower, upper);
    size_t np;
    plan_item *plan = search_plan(&g, NULL, ORC_PBC_NONE, &np);
    sctx s = {K_SINGLE, 0, cutoff * cutoff, &g, &g, NULL, NULL, NULL};
    orc_pairs *r = finish(run_plan(&s, plan, np, nthreads), &g);
    free(plan);
    grid_free(&g);
    return r;
}

/* distance_search.rs:928-954 */
orc_pairs *orc_search_single_pbc(REAL cutoff, const REAL *pos, const uint64_t *ids, size_t n, const orc_box *box,
                                 uint8_t pbc_dims, int nthreads) {
    grid_t g = {0};
    REAL ext[3];
    orc_lab_extents(box, ext);
    if (!grid_dims_from_extents(cutoff, ext, g.dims)) return NULL;
    grid_populate_pbc(&g, pos, ids, n, box, pbc_dims);
    size_t np;
    plan_item *plan = search_plan(&g, NULL, pbc_dims, &np);
    sctx s = {K_SINGLE, 1, cutoff * cutoff, &g, &g, box, NULL, NULL};
    orc_pairs *r = finish(run_plan(&s, plan, np, nthreads), &g);
    free(plan);
    grid_free(&g);
    return r;
}

/* distance_search.rs:659-698 */
orc_pairs *orc_search_double(REAL cutoff, const REAL *pos1, const uint64_t *ids1, size_t n1, const REAL *pos2,
                             const uint64_t *ids2, size_t n2, int nthreads) {
    REAL lower[3], upper[3];
    orc_bounding_box_double(cutoff, pos1, n1, pos2, n2, lower, upper);
    return search_two(K_DOUBLE, cutoff, pos1, ids1, n1, pos2, ids2, n2, lower, upper, NULL, 0, NULL, NULL, nthreads);
}

/* distance_search.rs:713-754 */
orc_pairs *orc_search_double_pbc(REAL cutoff, const REAL *pos1, const uint64_t *ids1, size_t n1, const REAL *pos2,
                                 const uint64_t *ids2, size_t n2, const orc_box *box, uint8_t pbc_dims,
                                 int nthreads) {
    return search_two(K_DOUBLE, cutoff, pos1, ids1, n1, pos2, ids2, n2, NULL, NULL, box, pbc_dims, NULL, NULL,
                      nthreads);
}

static REAL vdw_grid_cutoff(const REAL *vdw1, size_t n1, const REAL *vdw2, size_t n2) {   /* :781-783 */
    REAL m1 = vdw1[0], m2 = vdw2[0];
    for (size_t k = 1; k < n1; ++k) m1 = (vdw1[k] > m1 || m1 != m1) ? vdw1[k] : m1;
    for (size_t k = 1; k < n2; ++k) m2 = (vdw2[k] > m2 || m2 != m2) ? vdw2[k] : m2;
    return (m1 + m2) + (REAL)R_EPS;
}

/* distance_search.rs:767-814 — ids are LOCAL 0..n */
orc_pairs *orc_search_double_vdw(const REAL *pos1, size_t n1, const REAL *pos2, size_t n2, const REAL *vdw1,
                                 const REAL *vdw2, int nthreads) {
    if (n1 == 0 || n2 == 0) return empty_result();     /* the reference would panic on unwrap() */
    REAL cutoff = vdw_grid_cutoff(vdw1, n1, vdw2, n2);
    REAL lower[3], upper[3];
    orc_bounding_box_double(cutoff, pos1, n1, pos2, n2, lower, upper);
    return search_two(K_VDW, cutoff, pos1, NULL, n1, pos2, NULL, n2, lower, upper, NULL, 0, vdw1, vdw2, nthreads);
}

/* distance_search.rs:829-879 */
orc_pairs *orc_search_double_vdw_pbc(const REAL *pos1, size_t n1, const REAL *pos2, size_t n2, const REAL *vdw1,
                                     const REAL *vdw2, const orc_box *box, uint8_t pbc_dims, int nthreads) {
    if (n1 == 0 || n2 == 0) return empty_result();
    REAL cutoff = vdw_grid_cutoff(vdw1, n1, vdw2, n2);
    return search_two(K_VDW, cutoff, pos1, NULL, n1, pos2, NULL, n2, NULL, NULL, box, pbc_dims, vdw1, vdw2, nthreads);
}

/* distance_search.rs:519-558 */
orc_pairs *orc_search_within(REAL cutoff, const REAL *pos1, const uint64_t *ids1, size_t n1, const REAL *pos2,
                             const uint64_t *ids2, size_t n2, const REAL lower[3], const REAL upper[3],
                             int nthreads) {
    return search_two(K_WITHIN, cutoff, pos1, ids1, n1, pos2, ids2, n2, lower, upper, NULL, 0, NULL, NULL, nthreads);
}

/* distance_search.rs:560-598 */
orc_pairs *orc_search_within_pbc(REAL cutoff, const REAL *pos1, const uint64_t *ids1, size_t n1, const REAL *pos2,
                                 const uint64_t *ids2, size_t n2, const orc_box *box, uint8_t pbc_dims,
                                 int nthreads) {
    return search_two(K_WITHIN, cutoff, pos1, ids1, n1, pos2, ids2, n2, NULL, NULL, box, pbc_dims, NULL, NULL,
                      nthreads);
}

/* ------------------------------------------------------------------ brute force checker (not in the reference) */

orc_pairs *orc_brute_single(REAL cutoff, const REAL *pos, const uint64_t *ids, size_t n, const orc_box *box,
                            uint8_t pbc_dims) {
    pvec f = {0};
    f.with_jd = 1;
    REAL c2 = cutoff * cutoff;
    for (size_t i = 0; i + 1 < n; ++i)
        for (size_t j = i + 1; j < n; ++j) {
            REAL d2;
            if (box) d2 = orc_distance_squared(box, pos + 3 * i, pos + 3 * j, pbc_dims);
            else {
                REAL v[3] = {pos[3 * j] - pos[3 * i], pos[3 * j + 1] - pos[3 * i + 1], pos[3 * j + 2] - pos[3 * i + 2]};
                d2 = norm2_3(v);
            }
            if (d2 <= c2) pv_push(&f, ids ? ids[i] : i, ids ? ids[j] : j, R_SQRT(d2));
        }
    orc_pairs *out = (orc_pairs *)calloc(1, sizeof(orc_pairs));
    out->n = f.n;
    out->i = f.i ? f.i : (uint64_t *)malloc(8);
    out->j = f.j ? f.j : (uint64_t *)malloc(8);
    out->d = f.d ? f.d : (REAL *)malloc(8);
    return out;
}

orc_pairs *orc_brute_double(REAL cutoff, const REAL *pos1, const uint64_t *ids1, size_t n1, const REAL *pos2,
                            const uint64_t *ids2, size_t n2, const orc_box *box, uint8_t pbc_dims) {
    pvec f = {0};
    f.with_jd = 1;
    REAL c2 = cutoff * cutoff;
    for (size_t i = 0; i < n1; ++i)
        for (size_t j = 0; j < n2; ++j) {
            REAL d2;
            if (box) d2 = orc_distance_squared(box, pos1 + 3 * i, pos2 + 3 * j, pbc_dims);
            else {
                REAL v[3] = {pos2[3 * j] - pos1[3 * i], pos2[3 * j + 1] - pos1[3 * i + 1],
                             pos2[3 * j + 2] - pos1[3 * i + 2]};
                d2 = norm2_3(v);
            }
            if (d2 <= c2) pv_push(&f, ids1 ? ids1[i] : i, ids2 ? ids2[j] : j, R_SQRT(d2));
        }
    orc_pairs *out = (orc_pairs *)calloc(1, sizeof(orc_pairs));
    out->n = f.n;
    out->i = f.i ? f.i : (uint64_t *)malloc(8);
    out->j = f.j ? f.j : (uint64_t *)malloc(8);
    out->d = f.d ? f.d : (REAL *)malloc(8);
    return out;
}

/* ------------------------------------------------------------------ measure.rs */

#define POS(xyz, idx, k) ((xyz) + 3 * ((idx) ? (idx)[k] : (uint64_t)(k)))
#define MASS(mass, idx, k) ((mass)[(idx) ? (idx)[k] : (uint64_t)(k)])

/* measure.rs:22-36 */
void orc_min_max(const REAL *xyz, const uint64_t *idx, size_t n, REAL lower[3], REAL upper[3]) {
    for (int d = 0; d < 3; ++d) { lower[d] = (REAL)R_MAXVAL; upper[d] = -(REAL)R_MAXVAL; }
    for (size_t k = 0; k < n; ++k) {
        const REAL *p = POS(xyz, idx, k);
        for (int d = 0; d < 3; ++d) {
            if (p[d] < lower[d]) lower[d] = p[d];
            if (p[d] > upper[d]) upper[d] = p[d];
        }
    }
}

/* measure.rs:39-47 */
void orc_center_of_geometry(const REAL *xyz, const uint64_t *idx, size_t n, REAL out[3]) {
    REAL cog[3] = {0, 0, 0};
    for (size_t k = 0; k < n; ++k) {
        const REAL *p = POS(xyz, idx, k);
        cog[0] += p[0]; cog[1] += p[1]; cog[2] += p[2];
    }
    REAL nn = (REAL)n;
    out[0] = cog[0] / nn; out[1] = cog[1] / nn; out[2] = cog[2] / nn;
}

/* measure.rs:60-75 */
int orc_center_of_mass(const REAL *xyz, const uint64_t *idx, size_t n, const REAL *mass, REAL out[3]) {
    REAL cm[3] = {0, 0, 0}, m_tot = 0;
    for (size_t k = 0; k < n; ++k) {
        const REAL *p = POS(xyz, idx, k);
        REAL m = MASS(mass, idx, k);
        cm[0] += p[0] * m; cm[1] += p[1] * m; cm[2] += p[2] * m;
        m_tot += m;
    }
    if (m_tot == 0) return ORC_ERR_ZERO_MASS;
    out[0] = cm[0] / m_tot; out[1] = cm[1] / m_tot; out[2] = cm[2] / m_tot;
    return ORC_OK;
}

/* measure.rs:142-168 */
int orc_center_of_geometry_pbc_dims(const REAL *xyz, const uint64_t *idx, size_t n, const orc_box *b, uint8_t dims,
                                    REAL out[3]) {
    if (!b) return ORC_ERR_NO_PBC;
    const REAL *p0 = POS(xyz, idx, 0);
    REAL cm[3] = {p0[0], p0[1], p0[2]};
    for (size_t k = 1; k < n; ++k) {
        REAL im[3];
        orc_closest_image_dims(b, POS(xyz, idx, k), p0, dims, im);
        cm[0] += im[0]; cm[1] += im[1]; cm[2] += im[2];
    }
    REAL nn = (REAL)n;
    out[0] = cm[0] / nn; out[1] = cm[1] / nn; out[2] = cm[2] / nn;
    return ORC_OK;
}

/* measure.rs:172-220 — cm seeded with the UNWEIGHTED first position, mass with m0 */
int orc_center_of_mass_pbc_dims(const REAL *xyz, const uint64_t *idx, size_t n, const REAL *mass, const orc_box *b,
                                uint8_t dims, REAL out[3]) {
    if (!b) return ORC_ERR_NO_PBC;
    const REAL *p0 = POS(xyz, idx, 0);
    REAL m_tot = MASS(mass, idx, 0);
    REAL cm[3] = {p0[0], p0[1], p0[2]};
    for (size_t k = 1; k < n; ++k) {
        REAL im[3];
        REAL m = MASS(mass, idx, k);
        orc_closest_image_dims(b, POS(xyz, idx, k), p0, dims, im);
        cm[0] += im[0] * m; cm[1] += im[1] * m; cm[2] += im[2] * m;
        m_tot += m;
    }
    if (m_tot == 0) return ORC_ERR_ZERO_MASS;
    out[0] = cm[0] / m_tot; out[1] = cm[1] / m_tot; out[2] = cm[2] / m_tot;
    return ORC_OK;
}

/* displacement from the centre: plain (measure.rs:84) or shortest_vector (:229) */
static inline void disp(const REAL *p, const REAL c[3], const orc_box *b, REAL d[3]) {
    REAL v[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
    if (b) orc_shortest_vector_dims(b, v, ORC_PBC_FULL, d);
    else { d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; }
}

/* measure.rs:561-570 */
static REAL do_gyration(const REAL *xyz, const uint64_t *idx, size_t n, const REAL *mass, const REAL c[3],
                        const orc_box *b) {
    REAL sd = 0, sm = 0;
    for (size_t k = 0; k < n; ++k) {
        REAL d[3];
        disp(POS(xyz, idx, k), c, b, d);
        REAL m = MASS(mass, idx, k);
        sd += norm2_3(d) * m;
        sm += m;
    }
    return R_SQRT(sd / sm);
}

/* measure.rs:78-87 */
int orc_gyration(const REAL *xyz, const uint64_t *idx, size_t n, const REAL *mass, REAL *out) {
    REAL c[3];
    int rc = orc_center_of_mass(xyz, idx, n, mass, c);
    if (rc) return rc;
    *out = do_gyration(xyz, idx, n, mass, c, NULL);
    return ORC_OK;
}

/* measure.rs:222-232 */
int orc_gyration_pbc(const REAL *xyz, const uint64_t *idx, size_t n, const REAL *mass, const orc_box *b, REAL *out) {
    REAL c[3];
    int rc = orc_center_of_mass_pbc_dims(xyz, idx, n, mass, b, ORC_PBC_FULL, c);
    if (rc) return rc;
    *out = do_gyration(xyz, idx, n, mass, c, b);
    return ORC_OK;
}

/* cyclic Jacobi for a symmetric 3x3 in double; returns eigenvalues w[] and eigenvectors as
 * columns of V (column-major).  Stands in for nalgebra SymmetricEigen (measure.rs:592): any
 * correct symmetric eigensolver agrees to rounding; eigenvector signs are not pinned. */
static void jacobi_eig3(const double A_in[9], double w[3], double V[9]) {
    double A[9];
    memcpy(A, A_in, sizeof A);
    for (int i = 0; i < 9; ++i) V[i] = 0;
    V[0] = V[4] = V[8] = 1;
#define AA(r, c) A[(c) * 3 + (r)]
#define VV(r, c) V[(c) * 3 + (r)]
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = AA(0, 1) * AA(0, 1) + AA(0, 2) * AA(0, 2) + AA(1, 2) * AA(1, 2);
        double diag = AA(0, 0) * AA(0, 0) + AA(1, 1) * AA(1, 1) + AA(2, 2) * AA(2, 2);
        if (off <= 1e-300 || off <= 1e-34 * diag) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (AA(p, q) == 0) continue;
                double theta = (AA(q, q) - AA(p, p)) / (2.0 * AA(p, q));
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {          /* A <- A J */
                    double akp = AA(k, p), akq = AA(k, q);
                    AA(k, p) = c * akp - s * akq;
                    AA(k, q) = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {          /* A <- J^T A */
                    double apk = AA(p, k), aqk = AA(q, k);
                    AA(p, k) = c * apk - s * aqk;
                    AA(q, k) = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = VV(k, p), vkq = VV(k, q);
                    VV(k, p) = c * vkp - s * vkq;
                    VV(k, q) = s * vkp + c * vkq;
                }
            }
    }
    w[0] = AA(0, 0); w[1] = AA(1, 1); w[2] = AA(2, 2);
#undef AA
#undef VV
}

/* measure.rs:573-590: tensor accumulation */
static void inertia_tensor(const REAL *xyz, const uint64_t *idx, size_t n, const REAL *mass, const REAL c[3],
                           const orc_box *b, REAL tens[9]) {
    for (int i = 0; i < 9; ++i) tens[i] = 0;
    for (size_t k = 0; k < n; ++k) {
        REAL d[3];
        disp(POS(xyz, idx, k), c, b, d);
        REAL m = MASS(mass, idx, k);
        M(tens, 0, 0) += m * (d[1] * d[1] + d[2] * d[2]);
        M(tens, 1, 1) += m * (d[0] * d[0] + d[2] * d[2]);
        M(tens, 2, 2) += m * (d[0] * d[0] + d[1] * d[1]);
        M(tens, 0, 1) -= m * d[0] * d[1];
        M(tens, 0, 2) -= m * d[0] * d[2];
        M(tens, 1, 2) -= m * d[1] * d[2];
    }
    M(tens, 1, 0) = M(tens, 0, 1);
    M(tens, 2, 0) = M(tens, 0, 2);
    M(tens, 2, 1) = M(tens, 1, 2);
}

/* measure.rs:592-610 */
static void do_inertia(const REAL tens[9], REAL moments[3], REAL axes[9]) {
    double A[9], w[3], V[9];
    for (int i = 0; i < 9; ++i) A[i] = (double)tens[i];
    jacobi_eig3(A, w, V);
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a)
        for (int bq = a + 1; bq < 3; ++bq)
            if (w[ord[bq]] < w[ord[a]]) { int t = ord[a]; ord[a] = ord[bq]; ord[bq] = t; }
    moments[0] = (REAL)w[ord[0]]; moments[1] = (REAL)w[ord[1]]; moments[2] = (REAL)w[ord[2]];
    REAL c0[3], c1[3], c2[3], e0[3], e1[3];
    for (int r = 0; r < 3; ++r) { e0[r] = (REAL)V[ord[0] * 3 + r]; e1[r] = (REAL)V[ord[1] * 3 + r]; }
    normalize3(e0, c0);
    normalize3(e1, c1);
    cross3(c0, c1, c2);
    for (int r = 0; r < 3; ++r) { M(axes, r, 0) = c0[r]; M(axes, r, 1) = c1[r]; M(axes, r, 2) = c2[r]; }
}

int orc_inertia_tensor(const REAL *xyz, const uint64_t *idx, size_t n, const REAL *mass, const orc_box *b,
                       REAL tens9[9]) {
    REAL c[3];
    int rc = b ? orc_center_of_mass_pbc_dims(xyz, idx, n, mass, b, ORC_PBC_FULL, c)
               : orc_center_of_mass(xyz, idx, n, mass, c);
    if (rc) return rc;
    inertia_tensor(xyz, idx, n, mass, c, b, tens9);
    return ORC_OK;
}

/* measure.rs:90-99 */
int orc_inertia(const REAL *xyz, const uint64_t *idx, size_t n, const REAL *mass, REAL moments[3], REAL axes9[9]) {
    REAL tens[9];
    int rc = orc_inertia_tensor(xyz, idx, n, mass, NULL, tens);
    if (rc) return rc;
    do_inertia(tens, moments, axes9);
    return ORC_OK;
}

/* measure.rs:234-244 */
int orc_inertia_pbc(const REAL *xyz, const uint64_t *idx, size_t n, const REAL *mass, const orc_box *b,
                    REAL moments[3], REAL axes9[9]) {
    if (!b) return ORC_ERR_NO_PBC;
    REAL tens[9];
    int rc = orc_inertia_tensor(xyz, idx, n, mass, b, tens);
    if (rc) return rc;
    do_inertia(tens, moments, axes9);
    return ORC_OK;
}

/* measure.rs:485-504 */
int orc_rmsd(const REAL *xyz1, const uint64_t *idx1, size_t n1, const REAL *xyz2, const uint64_t *idx2, size_t n2,
             REAL *out) {
    if (n1 != n2) return ORC_ERR_SIZES;
    REAL res = 0;
    for (size_t k = 0; k < n1; ++k) {
        const REAL *p1 = POS(xyz1, idx1, k), *p2 = POS(xyz2, idx2, k);
        REAL v[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
        res += norm2_3(v);
    }
    *out = R_SQRT(res / (REAL)n1);
    return ORC_OK;
}

/* measure.rs:538-558 */
int orc_rmsd_mw(const REAL *xyz1, const uint64_t *idx1, size_t n1, const REAL *mass1, const REAL *xyz2,
                const uint64_t *idx2, size_t n2, REAL *out) {
    if (n1 != n2) return ORC_ERR_SIZES;
    REAL res = 0, m_tot = 0;
    for (size_t k = 0; k < n1; ++k) {
        const REAL *p1 = POS(xyz1, idx1, k), *p2 = POS(xyz2, idx2, k);
        REAL m = MASS(mass1, idx1, k);
        REAL v[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
        res += norm2_3(v) * m;
        m_tot += m;
    }
    if (m_tot == 0) return ORC_ERR_ZERO_MASS;
    *out = R_SQRT(res / m_tot);
    return ORC_OK;
}

/* One-sided (Hestenes) Jacobi SVD of a 3x3 in double: A = U S V^T, singular values sorted
 * descending.  Stands in for nalgebra::SVD::new(cov,true,true) (measure.rs:626); the Kabsch
 * rotation U diag(1,1,d) V^T is unique for a non-degenerate covariance, so any correct SVD
 * agrees to rounding. */
static void svd3(const double A_in[9], double U[9], double S[3], double V[9]) {
    double A[9];
    memcpy(A, A_in, sizeof A);
    for (int i = 0; i < 9; ++i) V[i] = 0;
    V[0] = V[4] = V[8] = 1;
#define AC(r, c) A[(c) * 3 + (r)]
#define VC(r, c) V[(c) * 3 + (r)]
    for (int sweep = 0; sweep < 64; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < 3; ++k) {
                    alpha += AC(k, p) * AC(k, p);
                    beta += AC(k, q) * AC(k, q);
                    gamma += AC(k, p) * AC(k, q);
                }
                if (gamma == 0 || fabs(gamma) <= 1e-17 * sqrt(alpha * beta)) continue;
                rotated = 1;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < 3; ++k) {
                    double akp = AC(k, p), akq = AC(k, q);
                    AC(k, p) = c * akp - s * akq;
                    AC(k, q) = s * akp + c * akq;
                    double vkp = VC(k, p), vkq = VC(k, q);
                    VC(k, p) = c * vkp - s * vkq;
                    VC(k, q) = s * vkp + c * vkq;
                }
            }
        if (!rotated) break;
    }
    double sv[3];
    for (int c = 0; c < 3; ++c) sv[c] = sqrt(AC(0, c) * AC(0, c) + AC(1, c) * AC(1, c) + AC(2, c) * AC(2, c));
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (sv[ord[b]] > sv[ord[a]]) { int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    double Vs[9];
    for (int c = 0; c < 3; ++c) {
        S[c] = sv[ord[c]];
        for (int r = 0; r < 3; ++r) {
            Vs[c * 3 + r] = VC(r, ord[c]);
            U[c * 3 + r] = S[c] > 0 ? AC(r, ord[c]) / S[c] : 0.0;
        }
    }
    memcpy(V, Vs, sizeof Vs);
    /* Rank two (a PLANAR selection: an aromatic ring, a leaflet of markers): the third column of A V is rounding noise,
     * so A V / S[2] is not a unit vector.  A true SVD (nalgebra's, measure.rs:626) returns an orthonormal U whatever the
     * rank, and U diag(1,1,d) V^T is still unique then: u2 = +-(u0 x u1), and the sign cancels against d (:631-641).
     * Rank one (collinear atoms) leaves the rotation about the line free: excluded from parity. */
    if (S[2] <= 1e-10 * S[0] || S[2] == 0) {
        if (S[1] > 0 && S[0] > 0) {
            double *u0 = U, *u1 = U + 3, *u2 = U + 6;
            u2[0] = u0[1] * u1[2] - u0[2] * u1[1];
            u2[1] = u0[2] * u1[0] - u0[0] * u1[2];
            u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
        }
    }
#undef AC
#undef VC
}

static double det3d(const double *m) {
    return m[0] * (m[4] * m[8] - m[7] * m[5]) - m[3] * (m[1] * m[8] - m[7] * m[2]) + m[6] * (m[1] * m[5] - m[4] * m[2]);
}

/* measure.rs:613-643.  q1/q2: centred coordinates, cov(r,c) += (q2[r]*q1[c])*m in REAL. */
static int rot_transform(const REAL *xyz1, const uint64_t *idx1, const REAL c1[3], const REAL *xyz2,
                         const uint64_t *idx2, const REAL c2[3], size_t n, const REAL *mass1, REAL R9[9]) {
    REAL cov[9] = {0};
    for (size_t k = 0; k < n; ++k) {
        const REAL *a = POS(xyz1, idx1, k), *b = POS(xyz2, idx2, k);
        REAL q1[3] = {a[0] - c1[0], a[1] - c1[1], a[2] - c1[2]};
        REAL q2[3] = {b[0] - c2[0], b[1] - c2[1], b[2] - c2[2]};
        REAL m = MASS(mass1, idx1, k);
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 3; ++r) M(cov, r, c) += (q2[r] * q1[c]) * m;
    }
    double A[9], U[9], S[3], V[9];
    for (int i = 0; i < 9; ++i) {
        if (cov[i] != cov[i]) return ORC_ERR_SVD;
        A[i] = (double)cov[i];
    }
    svd3(A, U, S, V);
    /* u * v_t */
    double UVt[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += U[k * 3 + r] * V[k * 3 + c];
            UVt[c * 3 + r] = s;
        }
    double d = det3d(UVt) < 0 ? -1.0 : 1.0;                 /* :631-635 */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = U[0 * 3 + r] * V[0 * 3 + c] + U[1 * 3 + r] * V[1 * 3 + c] + d * U[2 * 3 + r] * V[2 * 3 + c];
            M(R9, r, c) = (REAL)s;
        }
    return ORC_OK;
}

/* measure.rs:507-522: Translation(cm2) * rot * Translation(-cm1)  =>  t = cm2 + R*(-cm1) */
int orc_fit_transform(const REAL *xyz1, const uint64_t *idx1, size_t n1, const REAL *mass1, const REAL *xyz2,
                      const uint64_t *idx2, size_t n2, const REAL *mass2, REAL R9[9], REAL t3[3]) {
    REAL cm1[3], cm2[3];
    int rc = orc_center_of_mass(xyz1, idx1, n1, mass1, cm1);
    if (rc) return rc;
    rc = orc_center_of_mass(xyz2, idx2, n2, mass2, cm2);
    if (rc) return rc;
    size_t n = n1 < n2 ? n1 : n2;                             /* izip! stops at the shorter */
    rc = rot_transform(xyz1, idx1, cm1, xyz2, idx2, cm2, n, mass1, R9);
    if (rc) return rc;
    REAL neg[3] = {-cm1[0], -cm1[1], -cm1[2]}, rv[3];
    matvec(R9, neg, rv);
    t3[0] = cm2[0] + rv[0]; t3[1] = cm2[1] + rv[1]; t3[2] = cm2[2] + rv[2];
    return ORC_OK;
}

/* measure.rs:525-535 */
int orc_fit_transform_at_origin(const REAL *xyz1, const uint64_t *idx1, size_t n1, const REAL *mass1,
                                const REAL *xyz2, const uint64_t *idx2, size_t n2, REAL R9[9], REAL t3[3]) {
    REAL z[3] = {0, 0, 0};
    size_t n = n1 < n2 ? n1 : n2;
    int rc = rot_transform(xyz1, idx1, z, xyz2, idx2, z, n, mass1, R9);
    t3[0] = t3[1] = t3[2] = 0;
    return rc;
}

/* modify.rs:32-36: p <- R*p + t   (IsometryMatrix3 * Point3) */
void orc_apply_transform(REAL *xyz, const uint64_t *idx, size_t n, const REAL R9[9], const REAL t3[3]) {
    for (size_t k = 0; k < n; ++k) {
        REAL *p = (REAL *)POS(xyz, idx, k), r[3];
        matvec(R9, p, r);
        p[0] = r[0] + t3[0]; p[1] = r[1] + t3[1]; p[2] = r[2] + t3[2];
    }
}

/* modify.rs:16-23 */
void orc_translate(REAL *xyz, const uint64_t *idx, size_t n, const REAL shift[3]) {
    for (size_t k = 0; k < n; ++k) {
        REAL *p = (REAL *)POS(xyz, idx, k);
        p[0] += shift[0]; p[1] += shift[1]; p[2] += shift[2];
    }
}

/* modify.rs:40-54 */
int orc_unwrap_simple_dim(REAL *xyz, const uint64_t *idx, size_t n, const orc_box *b, uint8_t dims) {
    if (!b) return ORC_ERR_NO_PBC;
    if (n == 0) return ORC_OK;
    REAL p0[3];
    memcpy(p0, POS(xyz, idx, 0), sizeof p0);
    for (size_t k = 1; k < n; ++k) {
        REAL *p = (REAL *)POS(xyz, idx, k), o[3];
        orc_closest_image_dims(b, p, p0, dims, o);
        p[0] = o[0]; p[1] = o[1]; p[2] = o[2];
    }
    return ORC_OK;
}

static int cmp_u64(const void *a, const void *b) {
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* Modify::unwrap_connectivity_dim, modify.rs:72-131.  The search runs over the selection's positions with LOCAL ids 0..n
 * under PBC_FULL (:77-78); SearchConnectivity::from_iter (connectivity.rs:19-35) pushes j to i's list and i to j's in pair
 * order; the walk (:80-128) pops a centre, moves every not-yet-used neighbour to its closest image (over `dims`) relative
 * to the centre's CURRENT position, and pushes it.  A component starts from atom 0, later from the first unused atom
 * (find_position, :107); start atoms are not members of the returned selections (:97-98), empty selections are not
 * returned (:111-113,118-120).  select(&sel_vec) sorts.  Output: CSR of local indices (capacity n + 1 / n). */
int orc_unwrap_connectivity_dim(REAL *xyz, const uint64_t *idx, size_t n, const orc_box *b, REAL cutoff, uint8_t dims,
                                uint64_t *group_offsets, uint64_t *group_ids, size_t *ngroups, int nthreads) {
    if (!b) return ORC_ERR_NO_PBC;
    if (n == 0) return ORC_OK;
    REAL *pos = (REAL *)malloc(n * 3 * sizeof(REAL));
    for (size_t k = 0; k < n; ++k) memcpy(pos + 3 * k, POS(xyz, idx, k), 3 * sizeof(REAL));
    orc_pairs *pr = orc_search_single_pbc(cutoff, pos, NULL, n, b, 7, nthreads);
    free(pos);
    if (!pr) return ORC_ERR_NO_PBC;
    /* conn: i -> [j...] in push order */
    size_t *off = (size_t *)calloc(n + 2, sizeof(size_t));
    for (size_t p = 0; p < pr->n; ++p) { off[pr->i[p] + 1]++; off[pr->j[p] + 1]++; }
    for (size_t i = 0; i < n; ++i) off[i + 1] += off[i];
    uint64_t *adj = (uint64_t *)malloc((2 * pr->n + 1) * sizeof(uint64_t));
    size_t *cur = (size_t *)malloc((n + 1) * sizeof(size_t));
    memcpy(cur, off, (n + 1) * sizeof(size_t));
    for (size_t p = 0; p < pr->n; ++p) {
        adj[cur[pr->i[p]]++] = pr->j[p];
        adj[cur[pr->j[p]]++] = pr->i[p];
    }
    free(cur);
    orc_pairs_free(pr);
    unsigned char *used = (unsigned char *)calloc(n, 1);
    uint64_t *todo = (uint64_t *)malloc(n * sizeof(uint64_t)), *sel = (uint64_t *)malloc(n * sizeof(uint64_t));
    size_t ntodo = 0, nsel = 0, ng = 0, nids = 0;
    if (group_offsets) group_offsets[0] = 0;
    todo[ntodo++] = 0;
    used[0] = 1;
    for (;;) {
        while (ntodo) {
            const uint64_t c = todo[--ntodo];
            REAL p0[3];
            memcpy(p0, POS(xyz, idx, c), sizeof p0);
            for (size_t e = off[c]; e < off[c + 1]; ++e) {
                const uint64_t ind = adj[e];
                if (used[ind]) continue;
                REAL *p = (REAL *)POS(xyz, idx, ind), o[3];
                orc_closest_image_dims(b, p, p0, dims, o);
                p[0] = o[0]; p[1] = o[1]; p[2] = o[2];
                todo[ntodo++] = ind;
                used[ind] = 1;
                sel[nsel++] = ind;
            }
        }
        size_t i = 0;
        while (i < n && used[i]) ++i;
        const int last = i == n;
        if (!last) { todo[ntodo++] = i; used[i] = 1; }
        if (nsel) {
            /* select(&sel_vec): sorted */
            qsort(sel, nsel, sizeof(uint64_t), cmp_u64);
            for (size_t a = 0; a < nsel; ++a) { if (group_ids) group_ids[nids] = sel[a]; ++nids; }
            ++ng;
            if (group_offsets) group_offsets[ng] = nids;
            nsel = 0;
        }
        if (last) break;
    }
    if (ngroups) *ngroups = ng;
    free(used); free(todo); free(sel); free(off); free(adj);
    return ORC_OK;
}

/* measure.rs:270-422 */
int orc_lipid_tail_order(const REAL *xyz, const uint64_t *idx, size_t n, int order_type, const REAL *normals,
                         size_t n_normals, const uint8_t *bond_orders, size_t n_bonds, REAL *order) {
    if (n < 3) return ORC_ERR_LIPID_TAIL_TOO_SHORT;
    if (n_normals != 1 && n_normals != n - 2) return ORC_ERR_LIPID_NORMALS_COUNT;
    if (n_bonds != n - 1) return ORC_ERR_LIPID_BOND_ORDER_COUNT;
    for (size_t k = 0; k < n - 2; ++k) order[k] = 0;
#define NORMAL(k) (normals + 3 * (n_normals == 1 ? 0 : (k)))
#define SUB(a, b, o) do { (o)[0] = (a)[0] - (b)[0]; (o)[1] = (a)[1] - (b)[1]; (o)[2] = (a)[2] - (b)[2]; } while (0)
    if (order_type == 0) {
        for (size_t at = 1; at + 1 < n; ++at) {
            REAL v[3];
            SUB(POS(xyz, idx, at + 1), POS(xyz, idx, at - 1), v);
            REAL ang = angle3(v, NORMAL(at - 1));
            REAL c = R_COS(ang);
            order[at - 1] = (REAL)1.5 * (c * c) - (REAL)0.5;
        }
        return ORC_OK;
    }
    const REAL sqrt3 = R_SQRT((REAL)3.0);
    for (size_t i = 0; i + 2 < n; ++i) {
        if (bond_orders[i] == 1) {
            if (bond_orders[i + 1] == 1) {
                const REAL *p1 = POS(xyz, idx, i), *p2 = POS(xyz, idx, i + 1), *p3 = POS(xyz, idx, i + 2);
                REAL a[3], b[3], t[3], lz[3], lx[3], ly[3];
                SUB(p3, p1, t); normalize3(t, lz);
                SUB(p1, p2, a); SUB(p3, p2, b); cross3(a, b, t); normalize3(t, lx);
                cross3(lx, lz, ly);
                const REAL *nn = NORMAL(i);
                REAL cx = R_COS(angle3(lx, nn)), cy = R_COS(angle3(ly, nn));
                REAL sxx = (REAL)0.5 * ((REAL)3.0 * (cx * cx) - (REAL)1.0);
                REAL syy = (REAL)0.5 * ((REAL)3.0 * (cy * cy) - (REAL)1.0);
                order[i] = -((REAL)2.0 * sxx + syy) / (REAL)3.0;
            }
        } else {
            const REAL *p1 = POS(xyz, idx, i - 1), *p2 = POS(xyz, idx, i), *p3 = POS(xyz, idx, i + 1),
                       *p4 = POS(xyz, idx, i + 2);
            REAL a[3], b[3], t[3], lz[3], lx[3], ly[3];
            SUB(p1, p2, a); SUB(p3, p2, b);
            REAL a1 = (REAL)0.5 * ((REAL)R_PI - angle3(a, b));
            SUB(p2, p3, a); SUB(p4, p3, b);
            REAL a2 = (REAL)0.5 * ((REAL)R_PI - angle3(a, b));
            /* atom i */
            SUB(p3, p2, t); normalize3(t, lz);
            SUB(p1, p2, a); cross3(a, lz, t); normalize3(t, lx);
            cross3(lx, lz, ly);
            const REAL *n1 = NORMAL(i);
            REAL cy = R_COS(angle3(ly, n1)), cz = R_COS(angle3(lz, n1));
            REAL szz = (REAL)0.5 * ((REAL)3.0 * (cz * cz) - (REAL)1.0);
            REAL syy = (REAL)0.5 * ((REAL)3.0 * (cy * cy) - (REAL)1.0);
            REAL syz = (REAL)1.5 * cy * cz;
            if (order_type == 2) {
                REAL ca = R_COS(a1), sa = R_SIN(a1);
                order[i - 1] = -(((ca * ca) * syy + (sa * sa) * szz) - (REAL)2.0 * ca * sa * syz);
            } else {
                order[i - 1] = -((szz / (REAL)4.0 + (REAL)3.0 * syy / (REAL)4.0) - sqrt3 * syz / (REAL)2.0);
            }
            /* atom i+1 (same local_z) */
            SUB(p3, p4, a); cross3(a, lz, t); normalize3(t, lx);
            cross3(lx, lz, ly);
            const REAL *n2 = NORMAL(i + 1);
            cy = R_COS(angle3(ly, n2)); cz = R_COS(angle3(lz, n2));
            szz = (REAL)0.5 * ((REAL)3.0 * (cz * cz) - (REAL)1.0);
            syy = (REAL)0.5 * ((REAL)3.0 * (cy * cy) - (REAL)1.0);
            syz = (REAL)1.5 * cy * cz;
            if (order_type == 2) {
                REAL ca = R_COS(a2), sa = R_SIN(a2);
                order[i] = -(((ca * ca) * syy + (sa * sa) * szz) + (REAL)2.0 * ca * sa * syz);
            } else {
                order[i] = -((szz / (REAL)4.0 + (REAL)3.0 * syy / (REAL)4.0) + sqrt3 * syz / (REAL)2.0);
            }
        }
    }
#undef NORMAL
#undef SUB
    return ORC_OK;
}

/* molar_membrane/src/stats.rs:29-35 */
void orc_histogram_add(REAL minv, REAL maxv, size_t nbins, const REAL *vals, size_t nvals, REAL *bins) {
    int64_t n = (int64_t)nbins;
    for (size_t k = 0; k < nvals; ++k) {
        int64_t b = as_isize(R_FLOOR((REAL)n * (vals[k] - minv) / (maxv - minv)));
        if (b >= 0 && b < n) bins[b] += (REAL)1.0;
    }
}

/* ------------------------------------------------------------------ molar_membrane: Membrane::smooth
 *
 * One iteration of the surface smoothing (molar_membrane/src/lib.rs:661-812) for all lipids, with
 *   get_to_lab_transform          lipid_molecule.rs:190-196
 *   get_quad_coefs                lib.rs:844-863      (normal equations, nalgebra Cholesky + solve)
 *   VoronoiCell::new / add_point  molar/src/voronoi_cell.rs:62-211 (TOL 1e-10)
 *   compute_curvature_and_normal  lipid_molecule.rs:102-188
 *   z_surf / project_to_surf      lib.rs:865-879
 * PARITY UNPINNED: the reference holds no asserting test for this path (test_curvature_sphere only prints).
 * nalgebra's symmetric_eigen does not define the order or sign of the 2x2 eigenpairs; here: descending
 * eigenvalues, eigenvector with non-negative first non-zero component.
 */

/* nalgebra Cholesky::new (column-by-column, lower factor) then solve: L y = b, L^T x = y. */
static int chol6_solve(REAL m[36] /* column-major */, REAL b[6]) {
#define A(r, c) m[(c) * 6 + (r)]
    for (int j = 0; j < 6; ++j) {
        for (int k = 0; k < j; ++k) {
            const REAL factor = -A(j, k);
            for (int r = j; r < 6; ++r) A(r, j) = factor * A(r, k) + A(r, j);
        }
        const REAL diag = A(j, j);
        if (!(diag > 0)) return 0;           /* zero, negative or NaN pivot: not positive definite */
        const REAL denom = R_SQRT(diag);
        A(j, j) = denom;
        for (int r = j + 1; r < 6; ++r) A(r, j) /= denom;
    }
    for (int i = 0; i < 6; ++i) {            /* solve_lower_triangular_mut */
        const REAL coeff = b[i] / A(i, i);
        b[i] = coeff;
        for (int r = i + 1; r < 6; ++r) b[r] = (-coeff) * A(r, i) + b[r];
    }
    for (int i = 5; i >= 0; --i) {           /* ad_solve_lower_triangular_mut */
        REAL dot = 0;
        for (int r = i + 1; r < 6; ++r) dot += A(r, i) * b[r];
        b[i] = (b[i] - dot) / A(i, i);
    }
#undef A
    return 1;
}

static inline REAL z_surf(REAL x, REAL y, const REAL c[6]) {       /* lib.rs:870-879 */
    return ((((c[0] * x * x + c[1] * y * y) + c[2] * x * y) + c[3] * x) + c[4] * y) + c[5];
}

typedef struct { REAL x, y; int64_t next; int64_t id; } vvert;

static inline REAL vdist(const vvert *v, REAL lx, REAL ly, REAL r2) { return (lx * v->x + ly * v->y) - r2; }

/* voronoi_cell.rs:107-205; returns -1 if a loop would not terminate (the reference would hang) */
static int voro_add_point(vvert *vert, int64_t *nvert, int64_t *init, REAL px, REAL py, int64_t id) {
    const REAL TOL = (REAL)1e-10;
    const REAL lx = (REAL)0.5 * px, ly = (REAL)0.5 * py;
    const REAL r2 = lx * lx + ly * ly;
    int64_t cur = *init, guard = 0;
    REAL cur_d = vdist(&vert[cur], lx, ly, r2);
    while (cur_d >= TOL) {
        cur = vert[cur].next;
        cur_d = vdist(&vert[cur], lx, ly, r2);
        if (++guard > *nvert) return -1;
    }
    *init = cur;
    int64_t c1_in, c1_out, c2_in, c2_out;
    REAL c1_ind, c1_outd, c2_ind, c2_outd;
    for (;;) {
        const int64_t nx = vert[cur].next;
        if (nx == *init) return 0;
        const REAL nd = vdist(&vert[nx], lx, ly, r2);
        if (nd >= TOL) {
            c1_in = cur; c1_ind = cur_d; c1_out = nx; c1_outd = nd;
            cur = nx; cur_d = nd;
            break;
        }
        cur = nx; cur_d = nd;
    }
    guard = 0;
    for (;;) {
        const int64_t nx = vert[cur].next;
        const REAL nd = vdist(&vert[nx], lx, ly, r2);
        if (nd < TOL) {
            c2_out = cur; c2_outd = cur_d; c2_in = nx; c2_ind = nd;
            break;
        }
        cur = nx; cur_d = nd;
        if (++guard > *nvert) return -1;
    }
    {   /* cut 2 */
        const REAL frac = c2_outd / (R_ABS(c2_ind) + c2_outd);
        const REAL x = ((REAL)1.0 - frac) * vert[c2_out].x + frac * vert[c2_in].x;
        const REAL y = ((REAL)1.0 - frac) * vert[c2_out].y + frac * vert[c2_in].y;
        if (c1_out != c2_out) {
            vert[c2_out].x = x; vert[c2_out].y = y;
            vert[c1_out].next = c2_out;
        } else {
            vvert nv = {x, y, c2_in, vert[c2_out].id};
            vert[*nvert] = nv;
            vert[c1_out].next = *nvert;
            *nvert += 1;
        }
    }
    {   /* cut 1 */
        const REAL frac = c1_outd / (R_ABS(c1_ind) + c1_outd);
        const REAL x = ((REAL)1.0 - frac) * vert[c1_out].x + frac * vert[c1_in].x;
        const REAL y = ((REAL)1.0 - frac) * vert[c1_out].y + frac * vert[c1_in].y;
        vert[c1_out].x = x; vert[c1_out].y = y;
        vert[c1_out].id = id;
    }
    return 1;
}

/* 2x2 symmetric eigenproblem of [[a, b], [b, c]] (b = lower element, the one nalgebra reads) */
static void eig2_sym(REAL a, REAL b, REAL c, REAL w[2], REAL v[4] /* column-major 2x2 */) {
    const REAL half = (REAL)0.5;
    const REAL t = half * (a - c), m = half * (a + c);
    const REAL h = R_SQRT(t * t + b * b);
    w[0] = m + h; w[1] = m - h;
    REAL x, y;
    if (b == 0) {
        if (a >= c) { x = 1; y = 0; } else { x = 0; y = 1; }
    } else {
        /* (A - w1 I) v = 0 -> v = (b, w0 - a) or (w0 - c, b); take the better conditioned one */
        if (t >= 0) { x = t + h; y = b; } else { x = b; y = h - t; }
        const REAL n = R_SQRT(x * x + y * y);
        x /= n; y /= n;
        if (x < 0 || (x == 0 && y < 0)) { x = -x; y = -y; }
    }
    v[0] = x; v[1] = y;
    /* second eigenvector: perpendicular, same sign convention */
    REAL x2 = -y, y2 = x;
    if (x2 < 0 || (x2 == 0 && y2 < 0)) { x2 = -x2; y2 = -y2; }
    v[2] = x2; v[3] = y2;
}

int orc_membrane_smooth(const orc_box *box, size_t K, REAL *head, REAL *normals, uint8_t *valid,
                        const uint64_t *poff, const uint64_t *pids, REAL *coefs, REAL *mean_curv,
                        REAL *gauss_curv, REAL *princ_curvs, REAL *princ_dirs, REAL *area, uint32_t *nvert_out,
                        uint64_t *neib_ids, REAL *voro, REAL *fitted) {
    REAL *saved = (REAL *)malloc(sizeof(REAL) * 3 * (K ? K : 1));
    memcpy(saved, head, sizeof(REAL) * 3 * K);
    for (size_t i = 0; i < K; ++i) {
        if (!valid[i]) continue;
        const size_t p0 = poff[i], np = poff[i + 1] - poff[i], slot = p0 + 4 * i;
        const REAL *nrm = normals + 3 * i;
        REAL to_lab[9], to_local[9];
        {   /* lipid_molecule.rs:190-196 */
            const REAL ex[3] = {1, 0, 0};
            REAL c0[3], c1[3];
            cross3(nrm, ex, c0);
            cross3(nrm, c0, c1);
            for (int r = 0; r < 3; ++r) { to_lab[r] = c0[r]; to_lab[3 + r] = c1[r]; to_lab[6 + r] = -nrm[r]; }
        }
        if (!inverse3(to_lab, to_local)) { valid[i] = 0; continue; }
        REAL *lp = (REAL *)malloc(sizeof(REAL) * 3 * (np ? np : 1));
        for (size_t q = 0; q < np; ++q) {
            const REAL *s = saved + 3 * pids[p0 + q];
            REAL d[3] = {s[0] - saved[3 * i], s[1] - saved[3 * i + 1], s[2] - saved[3 * i + 2]}, sv[3];
            orc_shortest_vector_dims(box, d, ORC_PBC_FULL, sv);
            matvec(to_local, sv, lp + 3 * q);
        }
        REAL m[36], c[6] = {0, 0, 0, 0, 0, 0};
        memset(m, 0, sizeof m);
        for (size_t q = 0; q < np; ++q) {
            const REAL x = lp[3 * q], y = lp[3 * q + 1], z = lp[3 * q + 2];
            const REAL pw[6] = {x * x, y * y, x * y, x, y, (REAL)1.0};
            for (int cc = 0; cc < 6; ++cc)
                for (int r = 0; r < 6; ++r) m[cc * 6 + r] += pw[r] * pw[cc];
            for (int r = 0; r < 6; ++r) c[r] += pw[r] * z;
        }
        if (!chol6_solve(m, c)) { valid[i] = 0; free(lp); continue; }
        vvert *vert = (vvert *)malloc(sizeof(vvert) * (np + 4));
        const vvert w0 = {-10, -10, 1, -1}, w1 = {10, -10, 2, -2}, w2 = {10, 10, 3, -3}, w3 = {-10, 10, 0, -4};
        vert[0] = w0; vert[1] = w1; vert[2] = w2; vert[3] = w3;
        int64_t nv = 4, init = 0;
        int hang = 0;
        for (size_t q = 0; q < np && !hang; ++q)
            hang = voro_add_point(vert, &nv, &init, lp[3 * q], lp[3 * q + 1], (int64_t)pids[p0 + q]) < 0;
        /* direct neighbours (lib.rs:706-726) */
        uint32_t n_vert = 0, n_neib = 0;
        if (!hang) {
            int64_t cur = init;
            do {
                if (vert[cur].id >= 0) neib_ids[slot + n_neib++] = (uint64_t)vert[cur].id;
                n_vert++;
                cur = vert[cur].next;
            } while (cur != init);
        }
        if (hang || n_neib < n_vert) { valid[i] = 0; free(vert); free(lp); continue; }
        nvert_out[i] = n_vert;
        memcpy(coefs + 6 * i, c, sizeof c);
        {   /* lipid_molecule.rs:134-187 */
            const REAL a = c[0], b = c[1], cq = c[2], d = c[3], e = c[4];
            const REAL E = (REAL)1.0 + d * d, F = d * e, G = (REAL)1.0 + e * e;
            const REAL L = (REAL)2.0 * a, Mm = cq, N = (REAL)2.0 * b;
            const REAL Z = E * G - F * F;
            gauss_curv[i] = (L * N - Mm * Mm) / Z;
            mean_curv[i] = (REAL)0.5 * ((E * N - (REAL)2.0 * F * Mm) + G * L) / Z;
            const REAL g[3] = {d, e, (REAL)-1.0};
            REAL gn[3];
            normalize3(g, gn);
            matvec(to_lab, gn, normals + 3 * i);
            const REAL W00 = (E * L - F * Mm) / Z, W10 = (G * Mm - F * L) / Z, W11 = (G * N - F * Mm) / Z;
            REAL w[2], ev[4];
            eig2_sym(W00, W10, W11, w, ev);
            princ_curvs[2 * i] = w[0]; princ_curvs[2 * i + 1] = w[1];
            for (int k = 0; k < 2; ++k) {
                const REAL v3[3] = {ev[2 * k], ev[2 * k + 1], 0};
                matvec(to_lab, v3, princ_dirs + 6 * i + 3 * k);
            }
        }
        {   /* vertices projected to the surface, lab frame; triangle-fan area (lib.rs:731-752) */
            int64_t cur = init;
            for (uint32_t k = 0; k < n_vert; ++k) {
                const REAL pv[3] = {vert[cur].x, vert[cur].y, z_surf(vert[cur].x, vert[cur].y, c)};
                matvec(to_lab, pv, voro + 3 * (slot + k));
                cur = vert[cur].next;
            }
            REAL ar = 0;
            for (uint32_t k = 0; k < n_vert; ++k) {
                REAL cr[3];
                cross3(voro + 3 * (slot + k), voro + 3 * (slot + (k + 1) % n_vert), cr);
                ar += (REAL)0.5 * norm3(cr);
            }
            area[i] = ar;
        }
        for (size_t q = 0; q < np; ++q) {   /* fitted patch points (lib.rs:760-768) */
            const REAL dz[3] = {0, 0, z_surf(lp[3 * q], lp[3 * q + 1], c) - lp[3 * q + 2]};
            REAL t[3];
            matvec(to_lab, dz, t);
            const REAL *s = saved + 3 * pids[p0 + q];
            for (int r = 0; r < 3; ++r) fitted[3 * (p0 + q) + r] = s[r] + t[r];
        }
        free(vert); free(lp);
        if (R_ABS(c[5]) > (REAL)0.5) { valid[i] = 0; continue; }
        {
            const REAL dz[3] = {0, 0, c[5]};
            REAL t[3];
            matvec(to_lab, dz, t);
            for (int r = 0; r < 3; ++r) head[3 * i + r] += t[r];
        }
    }
    /* serial scatter-average (lib.rs:781-801) */
    REAL *sn = (REAL *)malloc(sizeof(REAL) * (K ? K : 1)), *sp = (REAL *)malloc(sizeof(REAL) * 3 * (K ? K : 1));
    for (size_t i = 0; i < K; ++i) sn[i] = 1;
    memcpy(sp, head, sizeof(REAL) * 3 * K);
    for (size_t i = 0; i < K; ++i) {
        if (!valid[i]) continue;
        for (uint64_t q = poff[i]; q < poff[i + 1]; ++q) {
            const uint64_t id = pids[q];
            sn[id] += (REAL)1.0;
            for (int r = 0; r < 3; ++r) sp[3 * id + r] += fitted[3 * q + r];
        }
    }
    for (size_t i = 0; i < K; ++i) {
        if (!valid[i]) continue;
        for (int r = 0; r < 3; ++r) head[3 * i + r] = sp[3 * i + r] / sn[i];
        const size_t slot = poff[i] + 4 * i;
        for (uint32_t k = 0; k < nvert_out[i]; ++k)
            for (int r = 0; r < 3; ++r) voro[3 * (slot + k) + r] += head[3 * i + r];
    }
    free(sn); free(sp); free(saved);
    return ORC_OK;
}
