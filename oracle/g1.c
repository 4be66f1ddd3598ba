/*
 * g1.c -- oracle: BLS12-381 G1 (y^2 = x^3 + 4 over Fq) and the multilinear-KZG prover side,
 * restated from multilinear_kzg/src/{multilinear_kzg,trusted_setup}.rs.
 * TEST INFRASTRUCTURE ONLY (see zkoracle.h).
 *
 * Group arithmetic in the reference is ark-ec 0.5.0 (un-vendored) [ext]; any correct
 * formulas produce the same GROUP ELEMENT, and results are only ever compared in
 * normalised affine form (a projective triple is not unique).  Jacobian coordinates,
 * a = 0: doubling "dbl-2009-l", addition "add-2007-bl" (Explicit-Formulas Database).
 * Commit / open are the reference's NAIVE algorithms: one double-and-add per term.
 */
#include "zk_internal.h"
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EL(base, i, n) ((base) + (size_t)(i) * (size_t)(n))
static int is_pow2(size_t x) { return x && !(x & (x - 1)); }

typedef struct { fe x, y, z; } jac;   /* z = 0 <=> infinity */

static const uint64_t GX[6] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL,
                               0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL};
static const uint64_t GY[6] = {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL,
                               0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};

static const field_t *FQ(void) { return orc_fld(ORC_FQ381); }
static const field_t *FR(void) { return orc_fld(ORC_FR381); }

static void jac_inf(jac *p) { fe_zero(&p->x); fe_zero(&p->y); fe_zero(&p->z); fe_one(FQ(), &p->x); fe_one(FQ(), &p->y); }
static int jac_is_inf(const jac *p) { return fe_is_zero(FQ(), &p->z); }

static void jac_from_affine(jac *p, const uint64_t *a12) {
    const field_t *F = FQ();
    fe_load(F, &p->x, a12);
    fe_load(F, &p->y, a12 + 6);
    if (fe_is_zero(F, &p->x) && fe_is_zero(F, &p->y)) jac_inf(p);
    else fe_one(F, &p->z);
}
static void jac_to_affine(uint64_t *a12, const jac *p) {
    const field_t *F = FQ();
    if (jac_is_inf(p)) { memset(a12, 0, 96); return; }
    fe zi, zi2, zi3, x, y;
    fe_inv(F, &zi, &p->z);
    fe_mul(F, &zi2, &zi, &zi);
    fe_mul(F, &zi3, &zi2, &zi);
    fe_mul(F, &x, &p->x, &zi2);
    fe_mul(F, &y, &p->y, &zi3);
    fe_store(F, a12, &x);
    fe_store(F, a12 + 6, &y);
}

static void jac_double(jac *o, const jac *p) {
    const field_t *F = FQ();
    if (jac_is_inf(p)) { *o = *p; return; }
    fe a, b, c, d, e, f, t, x3, y3, z3;
    fe_mul(F, &a, &p->x, &p->x);            /* A = X1^2 */
    fe_mul(F, &b, &p->y, &p->y);            /* B = Y1^2 */
    fe_mul(F, &c, &b, &b);                  /* C = B^2 */
    fe_add(F, &t, &p->x, &b);
    fe_mul(F, &t, &t, &t);
    fe_sub(F, &t, &t, &a);
    fe_sub(F, &t, &t, &c);
    fe_add(F, &d, &t, &t);                  /* D = 2((X1+B)^2 - A - C) */
    fe_add(F, &e, &a, &a);
    fe_add(F, &e, &e, &a);                  /* E = 3A */
    fe_mul(F, &f, &e, &e);                  /* F = E^2 */
    fe_sub(F, &x3, &f, &d);
    fe_sub(F, &x3, &x3, &d);                /* X3 = F - 2D */
    fe_sub(F, &t, &d, &x3);
    fe_mul(F, &y3, &e, &t);
    fe_add(F, &t, &c, &c);
    fe_add(F, &t, &t, &t);
    fe_add(F, &t, &t, &t);
    fe_sub(F, &y3, &y3, &t);                /* Y3 = E(D - X3) - 8C */
    fe_mul(F, &z3, &p->y, &p->z);
    fe_add(F, &z3, &z3, &z3);               /* Z3 = 2 Y1 Z1 */
    o->x = x3; o->y = y3; o->z = z3;
}

static void jac_add(jac *o, const jac *p, const jac *q) {
    const field_t *F = FQ();
    if (jac_is_inf(p)) { *o = *q; return; }
    if (jac_is_inf(q)) { *o = *p; return; }
    fe z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t, x3, y3, z3;
    fe_mul(F, &z1z1, &p->z, &p->z);
    fe_mul(F, &z2z2, &q->z, &q->z);
    fe_mul(F, &u1, &p->x, &z2z2);
    fe_mul(F, &u2, &q->x, &z1z1);
    fe_mul(F, &s1, &p->y, &q->z);
    fe_mul(F, &s1, &s1, &z2z2);
    fe_mul(F, &s2, &q->y, &p->z);
    fe_mul(F, &s2, &s2, &z1z1);
    if (fe_eq(F, &u1, &u2)) {
        if (fe_eq(F, &s1, &s2)) { jac_double(o, p); return; }   /* P == Q */
        jac_inf(o);                                             /* P == -Q */
        fe_zero(&o->z);
        return;
    }
    fe_sub(F, &h, &u2, &u1);
    fe_add(F, &i, &h, &h);
    fe_mul(F, &i, &i, &i);                  /* I = (2H)^2 */
    fe_mul(F, &j, &h, &i);                  /* J = H I */
    fe_sub(F, &r, &s2, &s1);
    fe_add(F, &r, &r, &r);                  /* r = 2(S2 - S1) */
    fe_mul(F, &v, &u1, &i);                 /* V = U1 I */
    fe_mul(F, &x3, &r, &r);
    fe_sub(F, &x3, &x3, &j);
    fe_sub(F, &x3, &x3, &v);
    fe_sub(F, &x3, &x3, &v);                /* X3 = r^2 - J - 2V */
    fe_sub(F, &t, &v, &x3);
    fe_mul(F, &y3, &r, &t);
    fe_mul(F, &t, &s1, &j);
    fe_add(F, &t, &t, &t);
    fe_sub(F, &y3, &y3, &t);                /* Y3 = r(V - X3) - 2 S1 J */
    fe_add(F, &z3, &p->z, &q->z);
    fe_mul(F, &z3, &z3, &z3);
    fe_sub(F, &z3, &z3, &z1z1);
    fe_sub(F, &z3, &z3, &z2z2);
    fe_mul(F, &z3, &z3, &h);                /* Z3 = ((Z1+Z2)^2 - Z1Z1 - Z2Z2) H */
    o->x = x3; o->y = y3; o->z = z3;
}

/* PrimeGroup::mul_bigint [ark-ec]: double-and-add over the canonical scalar, MSB first */
static void jac_mul_canon(jac *o, const jac *p, const uint64_t *k, int nlimbs) {
    jac acc;
    jac_inf(&acc);
    fe_zero(&acc.z);
    for (int i = 64 * nlimbs - 1; i >= 0; i--) {
        jac_double(&acc, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) jac_add(&acc, &acc, p);
    }
    *o = acc;
}
static void jac_mul_fr(jac *o, const jac *p, const uint64_t *scalar_mont) {
    fe s;
    uint64_t k[MAXL];
    fe_load(FR(), &s, scalar_mont);
    fe_to_canonical(FR(), k, &s);           /* into_bigint()  multilinear_kzg.rs:41 */
    jac_mul_canon(o, p, k, 4);
}

int orc_g1_generator(uint64_t *out12) {
    const field_t *F = FQ();
    fe x, y;
    fe_from_canonical(F, &x, GX);
    fe_from_canonical(F, &y, GY);
    fe_store(F, out12, &x);
    fe_store(F, out12 + 6, &y);
    return ORC_OK;
}
int orc_g1_is_on_curve(const uint64_t *p12) {
    const field_t *F = FQ();
    fe x, y, l, r, four;
    fe_load(F, &x, p12);
    fe_load(F, &y, p12 + 6);
    if (fe_is_zero(F, &x) && fe_is_zero(F, &y)) return 1;   /* infinity encoding */
    fe_mul(F, &l, &y, &y);
    fe_mul(F, &r, &x, &x);
    fe_mul(F, &r, &r, &x);
    fe_from_u64(F, &four, 4);
    fe_add(F, &r, &r, &four);
    return fe_eq(F, &l, &r);
}
int orc_g1_add(const uint64_t *p12, const uint64_t *q12, uint64_t *out12) {
    jac p, q, o;
    jac_from_affine(&p, p12);
    jac_from_affine(&q, q12);
    jac_add(&o, &p, &q);
    jac_to_affine(out12, &o);
    return ORC_OK;
}
int orc_g1_neg(const uint64_t *p12, uint64_t *out12) {
    const field_t *F = FQ();
    fe y;
    memcpy(out12, p12, 96);
    fe_load(F, &y, p12 + 6);
    fe_neg(F, &y, &y);
    fe_store(F, out12 + 6, &y);
    return ORC_OK;
}
int orc_g1_mul_fr(const uint64_t *p12, const uint64_t *scalar_fr, uint64_t *out12) {
    jac p, o;
    jac_from_affine(&p, p12);
    jac_mul_fr(&o, &p, scalar_fr);
    jac_to_affine(out12, &o);
    return ORC_OK;
}

/* compute_lagrange_basis, trusted_setup.rs:24-49 (O(n 2^n), exactly as written) */
int orc_kzg_lagrange_basis(const uint64_t *taus, size_t ntaus, uint64_t *out) {
    const field_t *F = FR();
    if (ntaus == 0) return ORC_E_ARG;                   /* :26 */
    size_t n = (size_t)1 << ntaus;
    fe one;
    fe_one(F, &one);
    for (size_t index = 0; index < n; index++) {        /* :32 */
        fe e = one;
        for (size_t i = 0; i < ntaus; i++) {
            size_t bit = (index >> (ntaus - 1 - i)) & 1;   /* :36 */
            fe tau, f;
            fe_load(F, &tau, EL(taus, i, 4));
            if (bit) f = tau; else fe_sub(F, &f, &one, &tau);
            fe_mul(F, &e, &e, &f);
        }
        fe_store(F, EL(out, index, 4), &e);
    }
    return ORC_OK;
}

/* compute_g1_powers_of_tau, trusted_setup.rs:51-60 */
int orc_kzg_setup_g1(const uint64_t *taus, size_t ntaus, uint64_t *out_points) {
    if (ntaus == 0) return ORC_E_ARG;
    size_t n = (size_t)1 << ntaus;
    uint64_t *basis = (uint64_t *)malloc(32 * n);
    if (!basis) return ORC_E_NOMEM;
    int rc = orc_kzg_lagrange_basis(taus, ntaus, basis);
    uint64_t g[12];
    orc_g1_generator(g);
    jac G, o;
    jac_from_affine(&G, g);
    for (size_t i = 0; i < n && rc == ORC_OK; i++) {
        jac_mul_fr(&o, &G, EL(basis, i, 4));
        jac_to_affine(EL(out_points, i, 12), &o);
    }
    free(basis);
    return rc;
}

/* sum_i [s_i] B_i  -- the reference's naive dot product, multilinear_kzg.rs:37-42 / :100-107 */
static void naive_msm(jac *acc, const uint64_t *scalars, const uint64_t *points, size_t n) {
    jac_inf(acc);
    fe_zero(&acc->z);
    for (size_t i = 0; i < n; i++) {
        jac b, t;
        jac_from_affine(&b, EL(points, i, 12));
        jac_mul_fr(&t, &b, EL(scalars, i, 4));
        jac_add(acc, acc, &t);
    }
}

int orc_kzg_commit(const uint64_t *values, size_t len, const uint64_t *g1_points, size_t npoints,
                   uint64_t *out12) {
    if (len != npoints) return ORC_E_KZG_LEN;           /* :29-33 */
    jac acc;
    naive_msm(&acc, values, g1_points, len);
    jac_to_affine(out12, &acc);
    return ORC_OK;
}

/* open_and_prove, multilinear_kzg.rs:50-126 */
int orc_kzg_open(const uint64_t *values, size_t len, const uint64_t *g1_points, size_t npoints,
                 const uint64_t *opening, size_t nopen, size_t n_g2, uint64_t *evaluation,
                 uint64_t *proofs) {
    const field_t *F = FR();
    if (!is_pow2(len)) return ORC_E_NOT_POW2;
    size_t nvars = 0;
    for (size_t x = len; x > 1; x >>= 1) nvars++;
    if (nvars != nopen) return ORC_E_KZG_LEN;           /* :55-59 */
    if (nopen != n_g2) return ORC_E_KZG_LEN;            /* :60-64 */
    int rc = orc_mle_evaluate(ORC_FR381, values, len, opening, nopen, evaluation);   /* :70 */
    if (rc != ORC_OK) return rc;
    fe v;
    fe_load(F, &v, evaluation);
    uint64_t *sub = (uint64_t *)malloc(32 * len);
    uint64_t *blown = (uint64_t *)malloc(32 * len);
    if (!sub || !blown) { free(sub); free(blown); return ORC_E_NOMEM; }
    for (size_t i = 0; i < len; i++) {                  /* :74-78 */
        fe x;
        fe_load(F, &x, EL(values, i, 4));
        fe_sub(F, &x, &x, &v);
        fe_store(F, EL(sub, i, 4), &x);
    }
    size_t cl = len;
    for (size_t i = 0; i < nopen && rc == ORC_OK; i++) {   /* :86 */
        size_t mid = cl / 2;                            /* compute_quotient_polynomial :165-179 */
        for (size_t k = 0; k < mid; k++) {
            fe lo, hi, q;
            fe_load(F, &lo, EL(sub, k, 4));
            fe_load(F, &hi, EL(sub, mid + k, 4));
            fe_sub(F, &q, &hi, &lo);
            fe_store(F, EL(blown, k, 4), &q);
        }
        /* blow_up :181-197 : (i+1) doublings by self-concatenation (expand_vec :199-209) */
        size_t bl = mid;
        for (size_t e = 0; e < i + 1; e++) {
            memcpy(EL(blown, bl, 4), blown, 32 * bl);
            bl *= 2;
        }
        if (bl != npoints) {
            /* zip() in :100-103 silently truncates; the sizes agree whenever the :29/:55 asserts held */
        }
        jac acc;
        naive_msm(&acc, blown, g1_points, bl < npoints ? bl : npoints);   /* :100-107 */
        jac_to_affine(EL(proofs, i, 12), &acc);
        fe x;
        fe_load(F, &x, EL(opening, i, 4));
        uint64_t *nx = (uint64_t *)malloc(32 * (mid ? mid : 1));
        if (!nx) { rc = ORC_E_NOMEM; break; }
        rc = mle_partial_evaluate(F, sub, cl, 0, &x, nx);   /* :113-117 */
        memcpy(sub, nx, 32 * mid);
        free(nx);
        cl = mid;
    }
    free(sub); free(blown);
    return rc;
}

int orc_kzg_quotients(const uint64_t *values, size_t len, const uint64_t *opening, size_t nopen,
                      uint64_t *out) {
    const field_t *F = FR();
    if (!is_pow2(len)) return ORC_E_NOT_POW2;
    uint64_t ev[4];
    int rc = orc_mle_evaluate(ORC_FR381, values, len, opening, nopen, ev);
    if (rc != ORC_OK) return rc;
    fe v;
    fe_load(F, &v, ev);
    uint64_t *sub = (uint64_t *)malloc(32 * len);
    if (!sub) return ORC_E_NOMEM;
    for (size_t i = 0; i < len; i++) {
        fe x;
        fe_load(F, &x, EL(values, i, 4));
        fe_sub(F, &x, &x, &v);
        fe_store(F, EL(sub, i, 4), &x);
    }
    size_t cl = len, off = 0;
    for (size_t i = 0; i < nopen && rc == ORC_OK; i++) {
        size_t mid = cl / 2;
        for (size_t k = 0; k < mid; k++) {
            fe lo, hi, q;
            fe_load(F, &lo, EL(sub, k, 4));
            fe_load(F, &hi, EL(sub, mid + k, 4));
            fe_sub(F, &q, &hi, &lo);
            fe_store(F, EL(out, off + k, 4), &q);
        }
        off += mid;
        fe x;
        fe_load(F, &x, EL(opening, i, 4));
        uint64_t *nx = (uint64_t *)malloc(32 * (mid ? mid : 1));
        rc = mle_partial_evaluate(F, sub, cl, 0, &x, nx);
        memcpy(sub, nx, 32 * mid);
        free(nx);
        cl = mid;
    }
    free(sub);
    return rc;
}

/* ---- CPU baseline legs (bench.py cpu_baseline only) ----------------------- */
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
/* the reference's fold as written: fresh Vec per call + the copy in new() (evaluation_form.rs:64,105,16) */
double orc_bench_fold(int field, const uint64_t *table, size_t len, const uint64_t *r, int reps) {
    const field_t *F = orc_fld(field);
    if (!F) return -1.0;
    fe v;
    fe_load(F, &v, r);
    size_t es = 8 * (size_t)F->n;
    double t0 = now_s();
    for (int k = 0; k < reps; k++) {
        uint64_t *res = (uint64_t *)malloc(es * (len / 2));
        mle_partial_evaluate(F, table, len, 0, &v, res);
        uint64_t *copy = (uint64_t *)malloc(es * (len / 2));
        memcpy(copy, res, es * (len / 2));
        free(res);
        free(copy);
    }
    return now_s() - t0;
}
/* best-effort multi-core variant: the same per-element arithmetic with the output
 * indices split over OpenMP threads (variable 0 only); returns seconds, *threads_used receives the team size */
double orc_bench_fold_mt(int field, const uint64_t *table, size_t len, const uint64_t *r, int reps, int *threads_used) {
    const field_t *F = orc_fld(field);
    if (!F) return -1.0;
    fe v;
    fe_load(F, &v, r);
    size_t n = (size_t)F->n, half = len / 2;
    uint64_t *res = (uint64_t *)malloc(8 * n * half);
    int used = 1;
    double t0 = now_s();
    for (int k = 0; k < reps; k++) {
#pragma omp parallel
        {
#ifdef _OPENMP
#pragma omp single
            used = omp_get_num_threads();
#endif
#pragma omp for schedule(static)
            for (long i = 0; i < (long)half; i++) {
                fe y1, y2, d, t, o;
                fe_load(F, &y1, table + (size_t)i * n);
                fe_load(F, &y2, table + ((size_t)i + half) * n);
                fe_sub(F, &d, &y2, &y1);
                fe_mul(F, &t, &v, &d);
                fe_add(F, &o, &y1, &t);
                fe_store(F, res + (size_t)i * n, &o);
            }
        }
    }
    double dt = now_s() - t0;
    free(res);
    if (threads_used) *threads_used = used;
    return dt;
}
double orc_bench_commit_naive(const uint64_t *values, size_t len, const uint64_t *g1_points) {
    uint64_t out[12];
    double t0 = now_s();
    orc_kzg_commit(values, len, g1_points, len, out);
    return now_s() - t0;
}

/* ---- best-effort CPU baseline for the MSM: the bucket method on all host cores -------------
 * NOT a restatement of reference code (the reference has no MSM routine, multilinear_kzg.rs:37-42 is a naive sum); it computes
 * the same group element as orc_kzg_commit (tests/test_oracle_kzg.py) and exists to be timed beside the GPU Pippenger.
 * Unsigned c-bit windows; work items = (window, slice of the terms), each with its own 2^c - 1 buckets and a running-sum
 * reduction; OpenMP over the items; per-window sums are combined by c doublings each, most significant window first. */
static unsigned window_digit(const uint64_t *k, int w, int c) {
    int bit = w * c, limb = bit / 64, off = bit % 64;
    uint64_t v = k[limb] >> off;
    if (off + c > 64 && limb + 1 < 4) v |= k[limb + 1] << (64 - off);
    return (unsigned)(v & (((uint64_t)1 << c) - 1));
}
int orc_msm_pippenger(const uint64_t *values, size_t len, const uint64_t *g1_points, int c, int slices, uint64_t *out12, int *threads_used) {
    if (c < 1 || c > 16 || slices < 1 || len == 0) return ORC_E_ARG;
    const int W = (255 + c - 1) / c;
    const size_t nb = ((size_t)1 << c) - 1;
    uint64_t *canon = (uint64_t *)malloc(32 * len);
    jac *partial = (jac *)malloc(sizeof(jac) * (size_t)W * (size_t)slices);
    if (!canon || !partial) { free(canon); free(partial); return ORC_E_ARG; }
    for (size_t i = 0; i < len; i++) {
        fe s;
        fe_load(FR(), &s, values + 4 * i);
        fe_to_canonical(FR(), canon + 4 * i, &s);                 /* into_bigint()  multilinear_kzg.rs:41 */
    }
    int used = 1;
#pragma omp parallel
    {
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#endif
        jac *buckets = (jac *)malloc(sizeof(jac) * nb);
#pragma omp for schedule(dynamic, 1)
        for (int item = 0; item < W * slices; item++) {
            const int w = item / slices, sl = item % slices;
            const size_t lo = len * (size_t)sl / (size_t)slices, hi = len * (size_t)(sl + 1) / (size_t)slices;
            for (size_t b = 0; b < nb; b++) { jac_inf(&buckets[b]); fe_zero(&buckets[b].z); }
            for (size_t i = lo; i < hi; i++) {
                unsigned dgt = window_digit(canon + 4 * i, w, c);
                if (!dgt) continue;
                jac p;
                jac_from_affine(&p, g1_points + 12 * i);
                jac_add(&buckets[dgt - 1], &buckets[dgt - 1], &p);
            }
            jac run, sum;                                          /* sum_b (b + 1) bucket[b] by the running sum */
            jac_inf(&run); fe_zero(&run.z);
            jac_inf(&sum); fe_zero(&sum.z);
            for (size_t b = nb; b-- > 0;) {
                jac_add(&run, &run, &buckets[b]);
                jac_add(&sum, &sum, &run);
            }
            partial[item] = sum;
        }
        free(buckets);
    }
    jac acc;
    jac_inf(&acc); fe_zero(&acc.z);
    for (int w = W - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) jac_double(&acc, &acc);
        for (int sl = 0; sl < slices; sl++) jac_add(&acc, &acc, &partial[(size_t)w * slices + sl]);
    }
    jac_to_affine(out12, &acc);
    free(canon);
    free(partial);
    if (threads_used) *threads_used = used;
    return ORC_OK;
}
double orc_bench_pippenger_mt(const uint64_t *values, size_t len, const uint64_t *g1_points, int c, int slices, uint64_t *out12, int *threads_used) {
    double t0 = now_s();
    if (orc_msm_pippenger(values, len, g1_points, c, slices, out12, threads_used) != ORC_OK) return -1.0;
    return now_s() - t0;
}
