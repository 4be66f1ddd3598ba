/*
 * mle.c -- oracle: MultilinearPolynomial, univariate interpolation, basic sumcheck,
 * composed polynomials and the GKR sumcheck, restated loop-for-loop from the reference.
 * TEST INFRASTRUCTURE ONLY (see zkoracle.h).
 */
#include "zk_internal.h"

#define GETF const field_t *F = orc_fld(field); if (!F) return ORC_E_ARG
#define EL(base, i) ((base) + (size_t)(i) * (size_t)F->n)

static int is_pow2(size_t x) { return x && !(x & (x - 1)); }
static unsigned ilog2(size_t x) { unsigned k = 0; while (x >>= 1) k++; return k; }

/* MultilinearPolynomial::new, evaluation_form.rs:12-18 (the assert only) */
int orc_mle_new_check(size_t len) { return is_pow2(len) ? ORC_OK : ORC_E_NOT_POW2; }

/* partial_evaluate, evaluation_form.rs:61-106 -- same index walk as the reference */
int mle_partial_evaluate(const field_t *F, const uint64_t *poly, size_t len, size_t var,
                         const fe *value, uint64_t *out) {
    size_t expected = len / 2;                        /* :63 */
    size_t i = 0, j = 0;
    while (i < expected) {                            /* :69 */
        fe y1, y2, d, t, o;
        fe_load(F, &y1, EL(poly, j));                 /* :70 */
        size_t nvars = ilog2(len);                    /* :74 */
        if (var + 1 > nvars) return ORC_E_RANGE;      /* :80 usize underflow */
        size_t power = nvars - 1 - var;               /* :80 */
        size_t k = j | ((size_t)1 << power);
        if (k >= len) return ORC_E_RANGE;             /* :82 index panic (non-pow2 input) */
        fe_load(F, &y2, EL(poly, k));                 /* :82 */
        fe_sub(F, &d, &y2, &y1);
        fe_mul(F, &t, value, &d);
        fe_add(F, &o, &y1, &t);                       /* :88-89  y1 + r*(y2 - y1) */
        fe_store(F, EL(out, i), &o);
        i += 1;
        j = ((j + 1) % ((size_t)1 << power) == 0) ? j + 1 + ((size_t)1 << power) : j + 1; /* :98 */
    }
    return is_pow2(expected) ? ORC_OK : ORC_E_NOT_POW2;   /* :105 -> new() :13 */
}

int orc_mle_partial_evaluate(int field, const uint64_t *poly, size_t len, size_t var,
                             const uint64_t *value, uint64_t *out) {
    GETF;
    fe v;
    fe_load(F, &v, value);
    return mle_partial_evaluate(F, poly, len, var, &v, out);
}

/* evaluate, evaluation_form.rs:21-33: clone, fold variable 0 `nvalues` times, take [0] */
int orc_mle_evaluate(int field, const uint64_t *poly, size_t len, const uint64_t *values,
                     size_t nvalues, uint64_t *out) {
    GETF;
    if (len == 0) return ORC_E_RANGE;
    uint64_t *cur = (uint64_t *)malloc(8 * (size_t)F->n * len);
    if (!cur) return ORC_E_NOMEM;
    memcpy(cur, poly, 8 * (size_t)F->n * len);       /* :22 clone */
    size_t cl = len;
    int rc = ORC_OK;
    for (size_t i = 0; i < nvalues; i++) {            /* :27 */
        uint64_t *nx = (uint64_t *)malloc(8 * (size_t)F->n * (cl / 2 ? cl / 2 : 1));
        if (!nx) { free(cur); return ORC_E_NOMEM; }
        fe v;
        fe_load(F, &v, EL(values, i));
        rc = mle_partial_evaluate(F, cur, cl, 0, &v, nx);   /* :28 */
        free(cur);
        cur = nx;
        cl /= 2;
        if (rc != ORC_OK) break;
    }
    if (rc == ORC_OK) memcpy(out, cur, 8 * (size_t)F->n);   /* :32 */
    free(cur);
    return rc;
}

/* convert_to_bytes, evaluation_form.rs:35-43: canonical, big-endian, per element */
int orc_mle_to_bytes(int field, const uint64_t *poly, size_t len, uint8_t *out) {
    GETF;
    for (size_t i = 0; i < len; i++) {
        fe x;
        fe_load(F, &x, EL(poly, i));
        fe_to_be_bytes(F, out + i * 8 * (size_t)F->n, &x);
    }
    return ORC_OK;
}

int orc_mle_scalar_mul(int field, const uint64_t *poly, size_t len, const uint64_t *s, uint64_t *out) {
    GETF;
    fe sc;
    fe_load(F, &sc, s);
    for (size_t i = 0; i < len; i++) {                /* :50-54 */
        fe x, o;
        fe_load(F, &x, EL(poly, i));
        fe_mul(F, &o, &x, &sc);
        fe_store(F, EL(out, i), &o);
    }
    return orc_mle_new_check(len);                    /* :56 */
}

int orc_mle_add(int field, const uint64_t *a, size_t la, const uint64_t *b, size_t lb, uint64_t *out) {
    GETF;
    if (la != lb) return ORC_E_LEN_MISMATCH;          /* :149 */
    for (size_t i = 0; i < la; i++) {
        fe x, y, o;
        fe_load(F, &x, EL(a, i));
        fe_load(F, &y, EL(b, i));
        fe_add(F, &o, &x, &y);
        fe_store(F, EL(out, i), &o);
    }
    return orc_mle_new_check(la);
}

static int tensor(int field, const uint64_t *wb, size_t lb, const uint64_t *wc, size_t lc,
                  uint64_t *out, int mul) {
    GETF;
    if (lb != lc) return ORC_E_LEN_MISMATCH;          /* :112 / :129 */
    size_t k = 0;
    for (size_t b = 0; b < lb; b++)                   /* :116 / :136  b-major */
        for (size_t c = 0; c < lc; c++) {
            fe x, y, o;
            fe_load(F, &x, EL(wb, b));
            fe_load(F, &y, EL(wc, c));
            if (mul) fe_mul(F, &o, &x, &y); else fe_add(F, &o, &x, &y);
            fe_store(F, EL(out, k++), &o);
        }
    return orc_mle_new_check(lb * lc);
}
int orc_mle_tensor_add(int field, const uint64_t *wb, size_t lb, const uint64_t *wc, size_t lc, uint64_t *out) {
    return tensor(field, wb, lb, wc, lc, out, 0);
}
int orc_mle_tensor_mul(int field, const uint64_t *wb, size_t lb, const uint64_t *wc, size_t lc, uint64_t *out) {
    return tensor(field, wb, lb, wc, lc, out, 1);
}

static void vec_sum(const field_t *F, const uint64_t *a, size_t n, fe *acc) {
    fe_zero(acc);
    for (size_t i = 0; i < n; i++) {
        fe x;
        fe_load(F, &x, EL(a, i));
        fe_add(F, acc, acc, &x);
    }
}
int orc_vec_sum(int field, const uint64_t *a, size_t n, uint64_t *out) {
    GETF;
    fe acc;
    vec_sum(F, a, n, &acc);
    fe_store(F, out, &acc);
    return ORC_OK;
}

/* ---- univariate: dense_univariate.rs ------------------------------------- */
static void uni_eval(const field_t *F, const fe *c, size_t n, const fe *x, fe *out) {
    fe result, power, t;                               /* :57-68 */
    fe_zero(&result);
    fe_one(F, &power);
    for (size_t i = 0; i < n; i++) {
        fe_mul(F, &t, &c[i], &power);
        fe_add(F, &result, &result, &t);
        fe_mul(F, &power, &power, x);
    }
    *out = result;
}
int orc_uni_evaluate(int field, const uint64_t *coeffs, size_t n, const uint64_t *x, uint64_t *out) {
    GETF;
    fe *c = (fe *)malloc(sizeof(fe) * (n ? n : 1));
    if (!c) return ORC_E_NOMEM;
    for (size_t i = 0; i < n; i++) fe_load(F, &c[i], EL(coeffs, i));
    fe xv, o;
    fe_load(F, &xv, x);
    uni_eval(F, c, n, &xv, &o);
    fe_store(F, out, &o);
    free(c);
    return ORC_OK;
}

/* lagrange_interpolate :74-98 with lagrange_basis :101-126; n points -> n coefficients */
static int lagrange(const field_t *F, const fe *xs, const fe *ys, size_t n, fe *out) {
    fe *num = (fe *)malloc(sizeof(fe) * (n + 1));
    fe *tmp = (fe *)malloc(sizeof(fe) * (n + 1));
    if (!num || !tmp) { free(num); free(tmp); return ORC_E_NOMEM; }
    for (size_t i = 0; i < n; i++) fe_zero(&out[i]);
    for (size_t idx = 0; idx < n; idx++) {
        size_t ln = 1;
        fe_one(F, &num[0]);                            /* :107 numerator = [1] */
        for (size_t k = 0; k < n; k++) {
            if (fe_eq(F, &xs[k], &xs[idx])) continue;  /* :112 compares VALUES */
            /* multiply_polynomials(numerator, [-x, 1]) :113 */
            fe nx;
            fe_neg(F, &nx, &xs[k]);
            for (size_t i = 0; i <= ln; i++) fe_zero(&tmp[i]);
            for (size_t i = 0; i < ln; i++) {
                fe t;
                fe_mul(F, &t, &num[i], &nx);
                fe_add(F, &tmp[i], &tmp[i], &t);
                fe_add(F, &tmp[i + 1], &tmp[i + 1], &num[i]);
            }
            ln += 1;
            memcpy(num, tmp, sizeof(fe) * ln);
        }
        fe den, deninv, s;
        uni_eval(F, num, ln, &xs[idx], &den);          /* :120-121 */
        fe_inv(F, &deninv, &den);
        fe_mul(F, &s, &ys[idx], &deninv);              /* :125 y / denominator */
        for (size_t i = 0; i < ln; i++) {
            fe t;
            fe_mul(F, &t, &s, &num[i]);
            fe_add(F, &out[i], &out[i], &t);           /* add_polynomials :161 */
        }
    }
    free(num); free(tmp);
    return ORC_OK;
}
int orc_uni_lagrange_interpolate(int field, const uint64_t *xs, const uint64_t *ys, size_t n, uint64_t *out) {
    GETF;
    fe *x = (fe *)malloc(sizeof(fe) * 3 * (n ? n : 1));
    if (!x) return ORC_E_NOMEM;
    fe *y = x + n, *o = x + 2 * n;
    for (size_t i = 0; i < n; i++) { fe_load(F, &x[i], EL(xs, i)); fe_load(F, &y[i], EL(ys, i)); }
    int rc = lagrange(F, x, y, n, o);
    for (size_t i = 0; i < n; i++) fe_store(F, EL(out, i), &o[i]);
    free(x);
    return rc;
}

/* ---- basic sumcheck: prover.rs / verifier.rs ------------------------------ */
int orc_split_and_sum(int field, const uint64_t *table, size_t len, uint64_t *out2) {
    GETF;
    size_t mid = len / 2;                              /* prover.rs:79 */
    fe l, r;
    vec_sum(F, table, mid, &l);
    vec_sum(F, EL(table, mid), len - mid, &r);
    fe_store(F, EL(out2, 0), &l);
    fe_store(F, EL(out2, 1), &r);
    return ORC_OK;
}

static void append_be(const field_t *F, orc_transcript *t, const fe *x) {
    uint8_t b[8 * MAXL];
    fe_to_be_bytes(F, b, x);
    orc_transcript_append(t, b, 8 * (size_t)F->n);
}
static void append_table_be(const field_t *F, orc_transcript *t, const uint64_t *tab, size_t len) {
    for (size_t i = 0; i < len; i++) {
        fe x;
        fe_load(F, &x, EL(tab, i));
        append_be(F, t, &x);
    }
}
static void challenge(const field_t *F, orc_transcript *t, fe *out) {
    uint8_t d[32];
    orc_transcript_sample(t, d);
    fe_from_le_bytes(F, out, d, 32);
}

int orc_sumcheck_basic_prove(int field, const uint64_t *table, size_t len, uint64_t *claimed_sum,
                             uint64_t *round_polys, uint64_t *challenges) {
    GETF;
    if (!is_pow2(len)) return ORC_E_NOT_POW2;          /* prover.rs:23 -> new() */
    size_t es = 8 * (size_t)F->n;
    fe claimed;
    vec_sum(F, table, len, &claimed);                  /* :28 */
    fe_store(F, claimed_sum, &claimed);
    orc_transcript *t = orc_transcript_new();
    if (!t) return ORC_E_NOMEM;
    append_table_be(F, t, table, len);                 /* :38-39 */
    append_be(F, t, &claimed);                         /* :40-41 */
    uint64_t *cur = (uint64_t *)malloc(es * len);
    if (!cur) { orc_transcript_free(t); return ORC_E_NOMEM; }
    memcpy(cur, table, es * len);                      /* :44 */
    size_t cl = len, nvars = ilog2(len);
    int rc = ORC_OK;
    for (size_t round = 0; round < nvars; round++) {   /* :46 */
        uint64_t *uni = EL(round_polys, 2 * round);
        orc_split_and_sum(field, cur, cl, uni);        /* :50 */
        append_table_be(F, t, uni, 2);                 /* :52-55 */
        fe r;
        challenge(F, t, &r);                           /* :58 */
        if (challenges) fe_store(F, EL(challenges, round), &r);
        uint64_t *nx = (uint64_t *)malloc(es * (cl / 2));
        if (!nx) { rc = ORC_E_NOMEM; break; }
        rc = mle_partial_evaluate(F, cur, cl, 0, &r, nx);   /* :61-63 */
        free(cur);
        cur = nx;
        cl /= 2;
        if (rc != ORC_OK) break;
    }
    free(cur);
    orc_transcript_free(t);
    return rc;
}

int orc_sumcheck_basic_verify(int field, const uint64_t *table, size_t len,
                              const uint64_t *claimed_sum, const uint64_t *round_polys,
                              size_t nrounds) {
    GETF;
    if (!is_pow2(len)) return ORC_E_NOT_POW2;
    if (nrounds != ilog2(len)) return 0;               /* verifier.rs:26-30 */
    fe cur;
    fe_load(F, &cur, claimed_sum);                     /* :32 */
    orc_transcript *t = orc_transcript_new();
    if (!t) return ORC_E_NOMEM;
    append_table_be(F, t, table, len);                 /* :34-35 */
    append_be(F, t, &cur);                             /* :36-37 */
    uint64_t *ch = (uint64_t *)malloc(8 * (size_t)F->n * (nrounds ? nrounds : 1));
    if (!ch) { orc_transcript_free(t); return ORC_E_NOMEM; }
    int ok = 1;
    for (size_t i = 0; i < nrounds && ok; i++) {       /* :47 */
        const uint64_t *uni = EL(round_polys, 2 * i);
        fe zero, one, e0, e1, s;
        fe_zero(&zero);
        fe_one(F, &one);
        uint64_t tmp[MAXL], o0[MAXL], o1[MAXL];
        fe_store(F, tmp, &zero);
        if (orc_mle_evaluate(field, uni, 2, tmp, 1, o0) != ORC_OK) { ok = 0; break; }   /* :51 */
        fe_store(F, tmp, &one);
        if (orc_mle_evaluate(field, uni, 2, tmp, 1, o1) != ORC_OK) { ok = 0; break; }   /* :52 */
        fe_load(F, &e0, o0);
        fe_load(F, &e1, o1);
        fe_add(F, &s, &e0, &e1);
        if (!fe_eq(F, &s, &cur)) { ok = 0; break; }    /* :53-56 */
        append_table_be(F, t, uni, 2);                 /* :58-59 */
        fe c;
        challenge(F, t, &c);                           /* :61 */
        fe_store(F, EL(ch, i), &c);
        if (orc_mle_evaluate(field, uni, 2, EL(ch, i), 1, o0) != ORC_OK) { ok = 0; break; } /* :64 */
        fe_load(F, &cur, o0);
    }
    if (ok) {
        uint64_t fin[MAXL];
        int rc = orc_mle_evaluate(field, table, len, ch, nrounds, fin);   /* :67 */
        fe f;
        fe_load(F, &f, fin);
        ok = (rc == ORC_OK) && fe_eq(F, &f, &cur);     /* :70 */
    }
    free(ch);
    orc_transcript_free(t);
    return ok;
}

/* ---- composed polynomials -------------------------------------------------- */
#define TAB(p, f) EL(tables, ((p) * nfac + (f)) * len)

int orc_sumpoly_evaluate(int field, const uint64_t *tables, size_t nprod, size_t nfac, size_t len,
                         const uint64_t *values, size_t nvalues, uint64_t *out) {
    GETF;
    fe result;
    fe_zero(&result);                                  /* sum_polynomial.rs:31 */
    for (size_t p = 0; p < nprod; p++) {
        fe prod;
        fe_one(F, &prod);                              /* product_polynomial.rs:27 */
        for (size_t f = 0; f < nfac; f++) {
            uint64_t o[MAXL];
            int rc = orc_mle_evaluate(field, TAB(p, f), len, values, nvalues, o);
            if (rc != ORC_OK) return rc;
            fe e;
            fe_load(F, &e, o);
            fe_mul(F, &prod, &prod, &e);
        }
        fe_add(F, &result, &result, &prod);
    }
    fe_store(F, out, &result);
    return ORC_OK;
}

/* add_polynomials_element_wise (sum_polynomial.rs:57-76) over
 * multiply_polynomials_element_wise (product_polynomial.rs:58-73) */
int orc_sumpoly_reduce(int field, const uint64_t *tables, size_t nprod, size_t nfac, size_t len,
                       uint64_t *out) {
    GETF;
    if (nprod < 2 || nfac < 2) return ORC_E_NEED_TWO;  /* :58-61 / :59-62 */
    for (size_t i = 0; i < len; i++) {
        fe acc;
        fe_zero(&acc);
        for (size_t p = 0; p < nprod; p++) {
            fe prod;
            fe_load(F, &prod, EL(TAB(p, 0), i));
            for (size_t f = 1; f < nfac; f++) {
                fe x;
                fe_load(F, &x, EL(TAB(p, f), i));
                fe_mul(F, &prod, &prod, &x);
            }
            if (p == 0) acc = prod; else fe_add(F, &acc, &acc, &prod);
        }
        fe_store(F, EL(out, i), &acc);
    }
    return orc_mle_new_check(len);
}

/* SumPolynomial::partial_evaluate (sum_polynomial.rs:40-53): every MLE folded at `var` */
static int sumpoly_fold(const field_t *F, const uint64_t *tables, size_t nprod, size_t nfac,
                        size_t len, const fe *v, uint64_t *out) {
    for (size_t k = 0; k < nprod * nfac; k++) {
        int rc = mle_partial_evaluate(F, EL(tables, k * len), len, 0, v, EL(out, k * (len / 2)));
        if (rc != ORC_OK) return rc;
    }
    return ORC_OK;
}

/* generate_round_univariate, sumcheck_gkr_protocol.rs:113-143 */
int orc_gkr_round_univariate(int field, const uint64_t *tables, size_t nprod, size_t nfac,
                             size_t len, uint64_t *out) {
    GETF;
    size_t es = 8 * (size_t)F->n, half = len / 2;
    size_t degree = nfac;                              /* :114 degree() = polynomials.len() */
    uint64_t *folded = (uint64_t *)malloc(es * nprod * nfac * (half ? half : 1));
    uint64_t *red = (uint64_t *)malloc(es * (half ? half : 1));
    if (!folded || !red) { free(folded); free(red); return ORC_E_NOMEM; }
    int rc = ORC_OK;
    for (size_t i = 0; i <= degree && rc == ORC_OK; i++) {   /* :127 */
        fe v, s;
        fe_from_u64(F, &v, (uint64_t)i);                     /* :128 */
        rc = sumpoly_fold(F, tables, nprod, nfac, len, &v, folded);   /* :129 */
        if (rc != ORC_OK) break;
        rc = orc_sumpoly_reduce(field, folded, nprod, nfac, half, red);  /* :133-134 */
        if (rc != ORC_OK) break;
        vec_sum(F, red, half, &s);                           /* :135-137 */
        fe_store(F, EL(out, i), &s);
    }
    free(folded); free(red);
    return rc;
}

static void append_le(const field_t *F, orc_transcript *t, const fe *x) {
    uint8_t b[8 * MAXL];
    fe_to_le_bytes(F, b, x);
    orc_transcript_append(t, b, 8 * (size_t)F->n);
}

/* prove, sumcheck_gkr_protocol.rs:24-67 */
int orc_sumcheck_gkr_prove(int field, const uint64_t *tables, size_t nprod, size_t nfac, size_t len,
                           const uint64_t *claimed_sum, orc_transcript *t, uint64_t *round_coeffs,
                           uint64_t *challenges) {
    GETF;
    if (!is_pow2(len)) return ORC_E_NOT_POW2;
    size_t es = 8 * (size_t)F->n, ntab = nprod * nfac, npts = nfac + 1;
    size_t nvars = ilog2(len);                         /* :29 */
    uint64_t *cur = (uint64_t *)malloc(es * ntab * len);
    if (!cur) return ORC_E_NOMEM;
    memcpy(cur, tables, es * ntab * len);              /* :33 clone */
    fe cs;
    fe_load(F, &cs, claimed_sum);
    append_be(F, t, &cs);                              /* :35 */
    fe *xs = (fe *)malloc(sizeof(fe) * 3 * npts);
    uint64_t *evals = (uint64_t *)malloc(es * npts);
    if (!xs || !evals) { free(cur); free(xs); free(evals); return ORC_E_NOMEM; }
    fe *ys = xs + npts, *co = xs + 2 * npts;
    size_t cl = len;
    int rc = ORC_OK;
    for (size_t round = 0; round < nvars; round++) {   /* :37 */
        rc = orc_gkr_round_univariate(field, cur, nprod, nfac, cl, evals);   /* :41 */
        if (rc != ORC_OK) break;
        for (size_t i = 0; i < npts; i++) {            /* :46-48 x = 0..=degree */
            fe_from_u64(F, &xs[i], (uint64_t)i);
            fe_load(F, &ys[i], EL(evals, i));
        }
        rc = lagrange(F, xs, ys, npts, co);            /* :49-50 */
        if (rc != ORC_OK) break;
        for (size_t i = 0; i < npts; i++) {
            append_le(F, t, &co[i]);                   /* :52 univariate_to_bytes: LITTLE endian */
            fe_store(F, EL(round_coeffs, round * npts + i), &co[i]);
        }
        fe r;
        challenge(F, t, &r);                           /* :55 */
        uint64_t *nx = (uint64_t *)malloc(es * ntab * (cl / 2));
        if (!nx) { rc = ORC_E_NOMEM; break; }
        rc = sumpoly_fold(F, cur, nprod, nfac, cl, &r, nx);   /* :57 */
        free(cur);
        cur = nx;
        cl /= 2;
        fe_store(F, EL(challenges, round), &r);        /* :59 */
        if (rc != ORC_OK) break;
    }
    free(cur); free(xs); free(evals);
    return rc;
}

/* verify, sumcheck_gkr_protocol.rs:69-105 */
int orc_sumcheck_gkr_verify(int field, const uint64_t *claimed_sum, const uint64_t *round_coeffs,
                            size_t nrounds, size_t ncoef, orc_transcript *t, uint64_t *challenges,
                            uint64_t *last_claimed_sum) {
    GETF;
    fe cur, zero, one;
    fe_load(F, &cur, claimed_sum);
    append_be(F, t, &cur);                             /* :73 */
    fe_zero(&zero);
    fe_one(F, &one);
    fe *c = (fe *)malloc(sizeof(fe) * (ncoef ? ncoef : 1));
    if (!c) return ORC_E_NOMEM;
    int ok = 1;
    for (size_t r = 0; r < nrounds; r++) {             /* :78 */
        for (size_t i = 0; i < ncoef; i++) fe_load(F, &c[i], EL(round_coeffs, r * ncoef + i));
        fe e0, e1, s;
        uni_eval(F, c, ncoef, &zero, &e0);             /* :81 */
        uni_eval(F, c, ncoef, &one, &e1);              /* :82 */
        fe_add(F, &s, &e0, &e1);
        if (!fe_eq(F, &s, &cur)) { ok = 0; break; }    /* :84-90 */
        for (size_t i = 0; i < ncoef; i++) append_le(F, t, &c[i]);   /* :92 */
        fe ch;
        challenge(F, t, &ch);                          /* :94 */
        uni_eval(F, c, ncoef, &ch, &cur);              /* :96 */
        fe_store(F, EL(challenges, r), &ch);
    }
    fe_store(F, last_claimed_sum, &cur);
    free(c);
    return ok;
}
