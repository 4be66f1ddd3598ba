/*
 * gkr.c -- oracle: layered circuit, wiring MLEs and the GKR prover / verifier, restated from
 * circuit/src/arithmetic_circuit.rs, gkr/src/utils.rs and gkr/src/gkr_protocol.rs.
 * TEST INFRASTRUCTURE ONLY (see zkoracle.h).
 */
#include "zk_internal.h"

#define GETF const field_t *F = orc_fld(field); if (!F) return ORC_E_ARG
#define EL(base, i) ((base) + (size_t)(i) * (size_t)F->n)
static int is_pow2(size_t x) { return x && !(x & (x - 1)); }

/* num_of_layer_variables, arithmetic_circuit.rs:166-178 */
size_t orc_num_layer_variables(size_t layer_index) {
    if (layer_index == 0) return 3;
    return layer_index + 2 * (layer_index + 1);
}
size_t orc_gkr_rounds(size_t layer_index) { return 2 * (layer_index + 1); }

/* format!("{:0>width$b}") :198-200 : at least `width` digits, never truncated, "0" for 0 */
static size_t padded_bits(size_t v, size_t width) {
    size_t nb = 1;
    while ((v >> nb) != 0) nb++;
    return nb > width ? nb : width;
}
/* convert_to_binary_and_to_decimal :180-196 : concatenate the three digit strings */
size_t orc_wiring_index(size_t layer_index, size_t a, size_t b, size_t c) {
    size_t wb = padded_bits(b, layer_index + 1), wc = padded_bits(c, layer_index + 1);
    return (((a << wb) | b) << wc) | c;
}

static size_t layer_len(const orc_gate *g, size_t ngates) {
    size_t mx = 0;                                      /* :73-78 max output index, default 0 */
    for (size_t i = 0; i < ngates; i++) if (g[i].out > mx) mx = g[i].out;
    return mx + 1;
}
size_t orc_circuit_eval_size(const orc_gate *gates, const size_t *gate_counts, size_t nlayers,
                             size_t ninputs) {
    size_t tot = ninputs, off = 0;
    for (size_t l = 0; l < nlayers; l++) { tot += layer_len(gates + off, gate_counts[l]); off += gate_counts[l]; }
    return tot;
}

/* Circuit::evaluate :65-109.  evals = layer 0 (output) ... layer nlayers (inputs), concatenated */
int orc_circuit_evaluate(int field, const orc_gate *gates, const size_t *gate_counts,
                         size_t nlayers, const uint64_t *inputs, size_t ninputs,
                         size_t *layer_sizes, uint64_t *evals) {
    GETF;
    size_t es = 8 * (size_t)F->n;
    /* offsets of each layer's gates and evaluations */
    size_t *goff = (size_t *)malloc(sizeof(size_t) * (nlayers + 1));
    size_t *eoff = (size_t *)malloc(sizeof(size_t) * (nlayers + 2));
    if (!goff || !eoff) { free(goff); free(eoff); return ORC_E_NOMEM; }
    goff[0] = 0;
    for (size_t l = 0; l < nlayers; l++) goff[l + 1] = goff[l] + gate_counts[l];
    eoff[0] = 0;
    for (size_t l = 0; l < nlayers; l++) {
        layer_sizes[l] = layer_len(gates + goff[l], gate_counts[l]);
        eoff[l + 1] = eoff[l] + layer_sizes[l];
    }
    layer_sizes[nlayers] = ninputs;
    memcpy(EL(evals, eoff[nlayers]), inputs, es * ninputs);          /* :69 */
    int rc = ORC_OK;
    for (size_t l = nlayers; l-- > 0 && rc == ORC_OK;) {             /* :72 layers.iter().rev() */
        const uint64_t *cur = EL(evals, eoff[l + 1]);
        size_t curlen = layer_sizes[l + 1];
        uint64_t *res = EL(evals, eoff[l]);
        memset(res, 0, es * layer_sizes[l]);                         /* :80 */
        for (size_t k = 0; k < gate_counts[l]; k++) {                /* :86 */
            const orc_gate *g = &gates[goff[l] + k];
            if (g->left >= curlen || g->right >= curlen) { rc = ORC_E_RANGE; break; }
            fe a, b, v, acc;
            fe_load(F, &a, EL(cur, g->left));
            fe_load(F, &b, EL(cur, g->right));
            if (g->op == 0) fe_add(F, &v, &a, &b); else fe_mul(F, &v, &a, &b);   /* :90-93 */
            fe_load(F, &acc, EL(res, g->out));
            fe_add(F, &acc, &acc, &v);                               /* :96 += */
            fe_store(F, EL(res, g->out), &acc);
        }
    }
    free(goff); free(eoff);
    return rc;
}

/* add_i_and_mul_i_mle :126-163 */
int orc_circuit_add_mul_mle(int field, const orc_gate *g, size_t ngates, size_t layer_index,
                            uint64_t *add_i, uint64_t *mul_i) {
    GETF;
    size_t n = (size_t)1 << orc_num_layer_variables(layer_index);
    memset(add_i, 0, 8 * (size_t)F->n * n);
    memset(mul_i, 0, 8 * (size_t)F->n * n);
    fe one;
    fe_one(F, &one);
    for (size_t k = 0; k < ngates; k++) {
        size_t pos = orc_wiring_index(layer_index, g[k].out, g[k].left, g[k].right);
        if (pos >= n) return ORC_E_RANGE;
        fe_store(F, EL(g[k].op == 0 ? add_i : mul_i, pos), &one);
    }
    return ORC_OK;
}

static void append_be(const field_t *F, orc_transcript *t, const fe *x) {
    uint8_t b[8 * MAXL];
    fe_to_be_bytes(F, b, x);
    orc_transcript_append(t, b, 8 * (size_t)F->n);
}
static void challenge(const field_t *F, orc_transcript *t, fe *out) {
    uint8_t d[32];
    orc_transcript_sample(t, d);
    fe_from_le_bytes(F, out, d, 32);
}

/* fold `tab` (len) by variable 0 with each of vals[0..k) in turn; result len >> k in *out (malloc) */
static int fold_chain(const field_t *F, const uint64_t *tab, size_t len, const fe *vals, size_t k,
                      uint64_t **out) {
    size_t es = 8 * (size_t)F->n;
    uint64_t *cur = (uint64_t *)malloc(es * len);
    if (!cur) return ORC_E_NOMEM;
    memcpy(cur, tab, es * len);
    for (size_t i = 0; i < k; i++) {
        uint64_t *nx = (uint64_t *)malloc(es * (len / 2 ? len / 2 : 1));
        if (!nx) { free(cur); return ORC_E_NOMEM; }
        int rc = mle_partial_evaluate(F, cur, len, 0, &vals[i], nx);
        free(cur);
        cur = nx;
        len /= 2;
        if (rc != ORC_OK) { free(cur); return rc; }
    }
    *out = cur;
    return ORC_OK;
}

/* compute_new_add_i_mul_i, utils.rs:23-68 : alpha*fold(x, rb) + beta*fold(x, rc) for x in {add, mul} */
static int new_add_mul(int field, const field_t *F, const fe *alpha, const fe *beta,
                       const uint64_t *add_abc, const uint64_t *mul_abc, size_t len, const fe *rb,
                       const fe *rc_, size_t k, uint64_t **new_add, uint64_t **new_mul) {
    if (k == 0) return ORC_E_RANGE;                     /* rb_values[0] :38 */
    const uint64_t *src[2] = {add_abc, mul_abc};
    uint64_t **dst[2] = {new_add, new_mul};
    size_t outlen = len >> k;
    uint64_t a_s[MAXL], b_s[MAXL];
    fe_store(F, a_s, alpha);
    fe_store(F, b_s, beta);
    for (int w = 0; w < 2; w++) {
        uint64_t *frb = NULL, *frc = NULL;
        int rc = fold_chain(F, src[w], len, rb, k, &frb);
        if (rc != ORC_OK) return rc;
        rc = fold_chain(F, src[w], len, rc_, k, &frc);
        if (rc != ORC_OK) { free(frb); return rc; }
        uint64_t *o = (uint64_t *)malloc(8 * (size_t)F->n * outlen);
        if (!o) { free(frb); free(frc); return ORC_E_NOMEM; }
        orc_mle_scalar_mul(field, frb, outlen, a_s, frb);            /* :58-59 */
        orc_mle_scalar_mul(field, frc, outlen, b_s, frc);
        rc = orc_mle_add(field, frb, outlen, frc, outlen, o);         /* :58,62 */
        free(frb); free(frc);
        if (rc != ORC_OK) { free(o); return rc; }
        *dst[w] = o;
    }
    return ORC_OK;
}

/* compute_fbc_polynomial, utils.rs:8-21 : tables [add_i_bc, W(b)+W(c), mul_i_bc, W(b)*W(c)] */
static int build_fbc(int field, const field_t *F, const uint64_t *add_bc, const uint64_t *mul_bc,
                     size_t bclen, const uint64_t *w, size_t wlen, uint64_t **out) {
    size_t es = 8 * (size_t)F->n;
    if (wlen * wlen != bclen) return ORC_E_NVARS;       /* ProductPolynomial::new product_polynomial.rs:16 */
    uint64_t *t = (uint64_t *)malloc(es * 4 * bclen);
    if (!t) return ORC_E_NOMEM;
    memcpy(EL(t, 0), add_bc, es * bclen);
    int rc = orc_mle_tensor_add(field, w, wlen, w, wlen, EL(t, bclen));
    memcpy(EL(t, 2 * bclen), mul_bc, es * bclen);
    if (rc == ORC_OK) rc = orc_mle_tensor_mul(field, w, wlen, w, wlen, EL(t, 3 * bclen));
    if (rc != ORC_OK) { free(t); return rc; }
    *out = t;
    return ORC_OK;
}

typedef struct { size_t *goff, *eoff, *lsz; uint64_t *evals; } circ_eval;
static void ce_free(circ_eval *c) { free(c->goff); free(c->eoff); free(c->lsz); free(c->evals); }
static int ce_make(int field, const field_t *F, const orc_gate *gates, const size_t *gate_counts,
                   size_t nlayers, const uint64_t *inputs, size_t ninputs, circ_eval *c) {
    memset(c, 0, sizeof *c);
    c->goff = (size_t *)calloc(nlayers + 1, sizeof(size_t));
    c->eoff = (size_t *)calloc(nlayers + 2, sizeof(size_t));
    c->lsz = (size_t *)calloc(nlayers + 1, sizeof(size_t));
    c->evals = (uint64_t *)malloc(8 * (size_t)F->n * orc_circuit_eval_size(gates, gate_counts, nlayers, ninputs));
    if (!c->goff || !c->eoff || !c->lsz || !c->evals) { ce_free(c); return ORC_E_NOMEM; }
    for (size_t l = 0; l < nlayers; l++) c->goff[l + 1] = c->goff[l] + gate_counts[l];
    int rc = orc_circuit_evaluate(field, gates, gate_counts, nlayers, inputs, ninputs, c->lsz, c->evals);
    for (size_t l = 0; l <= nlayers; l++) c->eoff[l + 1] = c->eoff[l] + c->lsz[l];
    if (rc != ORC_OK) ce_free(c);
    return rc;
}

/* gkr_protocol::prove, gkr_protocol.rs:26-143 */
int orc_gkr_prove(int field, const orc_gate *gates, const size_t *gate_counts, size_t nlayers,
                  const uint64_t *inputs, size_t ninputs, uint64_t *circuit_output,
                  size_t *output_len, uint64_t *claimed_sum, uint64_t *layer_claims,
                  uint64_t *coeffs, uint64_t *challenges, uint64_t *wb_evals, uint64_t *wc_evals) {
    GETF;
    size_t es = 8 * (size_t)F->n;
    circ_eval ce;
    int rc = ce_make(field, F, gates, gate_counts, nlayers, inputs, ninputs, &ce);   /* :27 */
    if (rc != ORC_OK) return rc;
    *output_len = ce.lsz[0];
    memcpy(circuit_output, ce.evals, es * ce.lsz[0]);

    orc_transcript *t = orc_transcript_new();
    fe alpha, beta, ra, claim;
    fe_zero(&alpha); fe_zero(&beta);
    fe *rb = NULL, *rcv = NULL;
    size_t nr = 0;

    /* w0, padded [x] -> [x, 0]  :39-47 */
    size_t w0len = ce.lsz[0];
    uint64_t *w0 = (uint64_t *)calloc((w0len == 1 ? 2 : w0len) * (size_t)F->n, 8);
    memcpy(w0, ce.evals, es * w0len);
    if (w0len == 1) w0len = 2;
    if (!is_pow2(w0len)) { rc = ORC_E_NOT_POW2; goto done; }
    {
        uint8_t *bytes = (uint8_t *)malloc(es * w0len);
        orc_mle_to_bytes(field, w0, w0len, bytes);
        orc_transcript_append(t, bytes, es * w0len);                 /* :49 */
        free(bytes);
    }
    challenge(F, t, &ra);                                            /* :50 */
    {
        uint64_t v[MAXL], o[MAXL];
        fe_store(F, v, &ra);
        rc = orc_mle_evaluate(field, w0, w0len, v, 1, o);            /* :51 */
        if (rc != ORC_OK) goto done;
        fe_load(F, &claim, o);
    }

    size_t coff = 0, choff = 0;
    for (size_t L = 0; L < nlayers; L++) {                           /* :57 */
        size_t abclen = (size_t)1 << orc_num_layer_variables(L);
        uint64_t *add_abc = (uint64_t *)malloc(es * abclen), *mul_abc = (uint64_t *)malloc(es * abclen);
        uint64_t *add_bc = NULL, *mul_bc = NULL, *fbc = NULL;
        size_t bclen = 0;
        rc = orc_circuit_add_mul_mle(field, gates + ce.goff[L], gate_counts[L], L, add_abc, mul_abc);   /* :58 */
        if (rc == ORC_OK) {
            if (L == 0) {                                            /* :60-72 */
                bclen = abclen / 2;
                add_bc = (uint64_t *)malloc(es * bclen);
                mul_bc = (uint64_t *)malloc(es * bclen);
                rc = mle_partial_evaluate(F, add_abc, abclen, 0, &ra, add_bc);
                if (rc == ORC_OK) rc = mle_partial_evaluate(F, mul_abc, abclen, 0, &ra, mul_bc);
            } else {                                                 /* :73-82 */
                bclen = abclen >> nr;
                rc = new_add_mul(field, F, &alpha, &beta, add_abc, mul_abc, abclen, rb, rcv, nr, &add_bc, &mul_bc);
            }
        }
        free(add_abc); free(mul_abc);
        const uint64_t *w = EL(ce.evals, ce.eoff[L + 1]);            /* :88-89 w_{i+1} */
        size_t wlen = ce.lsz[L + 1];
        if (rc == ORC_OK && !is_pow2(wlen)) rc = ORC_E_NOT_POW2;
        if (rc == ORC_OK) rc = build_fbc(field, F, add_bc, mul_bc, bclen, w, wlen, &fbc);   /* :95 */
        free(add_bc); free(mul_bc);
        if (rc != ORC_OK) { free(fbc); goto done; }
        size_t rounds = 0;
        for (size_t x = bclen; x > 1; x >>= 1) rounds++;
        uint64_t cs[MAXL];
        fe_store(F, cs, &claim);
        fe_store(F, EL(layer_claims, L), &claim);
        rc = orc_sumcheck_gkr_prove(field, fbc, 2, 2, bclen, cs, t, EL(coeffs, coff), EL(challenges, choff));   /* :99 */
        free(fbc);
        if (rc != ORC_OK) goto done;
        if (L < nlayers - 1) {                                       /* :109 */
            size_t mid = rounds / 2;                                 /* :120 */
            uint64_t o[MAXL];
            fe wbe, wce, ta, tb;
            rc = orc_mle_evaluate(field, w, wlen, EL(challenges, choff), mid, o);             /* utils.rs:78 */
            if (rc != ORC_OK) goto done;
            fe_load(F, &wbe, o);
            rc = orc_mle_evaluate(field, w, wlen, EL(challenges, choff + mid), rounds - mid, o);  /* :79 */
            if (rc != ORC_OK) goto done;
            fe_load(F, &wce, o);
            fe_store(F, EL(wb_evals, L), &wbe);                      /* :116-117 */
            fe_store(F, EL(wc_evals, L), &wce);
            free(rb); free(rcv);
            nr = mid;
            rb = (fe *)malloc(sizeof(fe) * (mid ? mid : 1));
            rcv = (fe *)malloc(sizeof(fe) * ((rounds - mid) ? (rounds - mid) : 1));
            for (size_t i = 0; i < mid; i++) fe_load(F, &rb[i], EL(challenges, choff + i));
            for (size_t i = 0; i < rounds - mid; i++) fe_load(F, &rcv[i], EL(challenges, choff + mid + i));
            append_be(F, t, &wbe);                                   /* :125 */
            challenge(F, t, &alpha);
            append_be(F, t, &wce);                                   /* :128 */
            challenge(F, t, &beta);
            fe_mul(F, &ta, &alpha, &wbe);
            fe_mul(F, &tb, &beta, &wce);
            fe_add(F, &claim, &ta, &tb);                             /* :132 */
        }
        coff += rounds * 3;
        choff += rounds;
    }
    fe_store(F, claimed_sum, &claim);
done:
    free(rb); free(rcv); free(w0);
    orc_transcript_free(t);
    ce_free(&ce);
    return rc;
}

/* expected claim: utils.rs:84-111 (layer 0) and :113-135 (folded) */
static int expected_claim(int field, const field_t *F, const uint64_t *add_bc, const uint64_t *mul_bc,
                          size_t bclen, const uint64_t *chal, size_t nch, const fe *wb, const fe *wc,
                          fe *out) {
    uint64_t o[MAXL];
    fe ar, mr, s, p, t1, t2;
    int rc = orc_mle_evaluate(field, add_bc, bclen, chal, nch, o);
    if (rc != ORC_OK) return rc;
    fe_load(F, &ar, o);
    rc = orc_mle_evaluate(field, mul_bc, bclen, chal, nch, o);
    if (rc != ORC_OK) return rc;
    fe_load(F, &mr, o);
    fe_add(F, &s, wb, wc);
    fe_mul(F, &p, wb, wc);
    fe_mul(F, &t1, &ar, &s);
    fe_mul(F, &t2, &mr, &p);
    fe_add(F, out, &t1, &t2);
    return ORC_OK;
}

/* gkr_protocol::verify, gkr_protocol.rs:146-236 */
int orc_gkr_verify(int field, const orc_gate *gates, const size_t *gate_counts, size_t nlayers,
                   const uint64_t *inputs, size_t ninputs, const uint64_t *circuit_output,
                   size_t output_len, const uint64_t *layer_claims, const uint64_t *coeffs,
                   const uint64_t *challenges_unused, const uint64_t *wb_evals, const uint64_t *wc_evals) {
    GETF;
    (void)challenges_unused;   /* the verifier re-derives challenges from its own transcript */
    size_t es = 8 * (size_t)F->n;
    orc_transcript *t = orc_transcript_new();
    fe alpha, beta, ra, claim;
    fe_zero(&alpha); fe_zero(&beta);
    size_t w0len = output_len == 1 ? 2 : output_len;                 /* :153-159 */
    uint64_t *w0 = (uint64_t *)calloc(w0len * (size_t)F->n, 8);
    memcpy(w0, circuit_output, es * output_len);
    int result = 0, rc;
    uint64_t *prev = NULL;
    size_t nprev = 0;
    if (!is_pow2(w0len)) { result = ORC_E_NOT_POW2; goto done; }
    {
        uint8_t *bytes = (uint8_t *)malloc(es * w0len);
        orc_mle_to_bytes(field, w0, w0len, bytes);
        orc_transcript_append(t, bytes, es * w0len);                 /* :161 */
        free(bytes);
        challenge(F, t, &ra);                                        /* :162 */
        uint64_t v[MAXL], o[MAXL];
        fe_store(F, v, &ra);
        rc = orc_mle_evaluate(field, w0, w0len, v, 1, o);            /* :164 */
        if (rc != ORC_OK) { result = rc; goto done; }
        fe_load(F, &claim, o);
    }
    size_t goff = 0, coff = 0;
    for (size_t L = 0; L < nlayers; L++) {                           /* :166 */
        fe lc;
        fe_load(F, &lc, EL(layer_claims, L));
        if (!fe_eq(F, &claim, &lc)) goto done;                       /* :167-169 */
        size_t rounds = orc_gkr_rounds(L);
        uint64_t *ch = (uint64_t *)malloc(es * rounds);
        uint64_t last[MAXL];
        int ok = orc_sumcheck_gkr_verify(field, EL(layer_claims, L), EL(coeffs, coff), rounds, 3, t, ch, last);   /* :172 */
        if (ok != 1) { free(ch); goto done; }                        /* :174-176 */
        fe wbe, wce;
        size_t mid = rounds / 2;
        if (L < nlayers - 1) {                                       /* :183-187 */
            fe_load(F, &wbe, EL(wb_evals, L));
            fe_load(F, &wce, EL(wc_evals, L));
        } else {                                                     /* :188-194 verifier's own inputs */
            uint64_t o[MAXL];
            if (!is_pow2(ninputs)) { free(ch); result = ORC_E_NOT_POW2; goto done; }
            rc = orc_mle_evaluate(field, inputs, ninputs, ch, mid, o);
            if (rc != ORC_OK) { free(ch); result = rc; goto done; }
            fe_load(F, &wbe, o);
            rc = orc_mle_evaluate(field, inputs, ninputs, EL(ch, mid), rounds - mid, o);
            if (rc != ORC_OK) { free(ch); result = rc; goto done; }
            fe_load(F, &wce, o);
        }
        size_t abclen = (size_t)1 << orc_num_layer_variables(L);
        uint64_t *add_abc = (uint64_t *)malloc(es * abclen), *mul_abc = (uint64_t *)malloc(es * abclen);
        uint64_t *add_bc = NULL, *mul_bc = NULL;
        size_t bclen;
        rc = orc_circuit_add_mul_mle(field, gates + goff, gate_counts[L], L, add_abc, mul_abc);
        if (rc == ORC_OK) {
            if (L == 0) {                                            /* :199-207 utils.rs:84 */
                bclen = abclen / 2;
                add_bc = (uint64_t *)malloc(es * bclen);
                mul_bc = (uint64_t *)malloc(es * bclen);
                rc = mle_partial_evaluate(F, add_abc, abclen, 0, &ra, add_bc);
                if (rc == ORC_OK) rc = mle_partial_evaluate(F, mul_abc, abclen, 0, &ra, mul_bc);
            } else {                                                 /* :208-219 utils.rs:113 */
                size_t k = nprev / 2;
                fe *prb = (fe *)malloc(sizeof(fe) * (k ? k : 1)), *prc = (fe *)malloc(sizeof(fe) * ((nprev - k) ? (nprev - k) : 1));
                for (size_t i = 0; i < k; i++) fe_load(F, &prb[i], EL(prev, i));
                for (size_t i = 0; i < nprev - k; i++) fe_load(F, &prc[i], EL(prev, k + i));
                bclen = abclen >> k;
                rc = new_add_mul(field, F, &alpha, &beta, add_abc, mul_abc, abclen, prb, prc, k, &add_bc, &mul_bc);
                free(prb); free(prc);
            }
        }
        free(add_abc); free(mul_abc);
        fe expect;
        if (rc == ORC_OK) rc = expected_claim(field, F, add_bc, mul_bc, bclen, ch, rounds, &wbe, &wce, &expect);
        free(add_bc); free(mul_bc);
        if (rc != ORC_OK) { free(ch); result = rc; goto done; }
        fe lastf;
        fe_load(F, &lastf, last);
        if (!fe_eq(F, &expect, &lastf)) { free(ch); goto done; }     /* :221-223 */
        free(prev);
        prev = ch;                                                   /* :225 */
        nprev = rounds;
        append_be(F, t, &wbe);                                       /* :227 */
        challenge(F, t, &alpha);
        append_be(F, t, &wce);                                       /* :230 */
        challenge(F, t, &beta);
        fe ta, tb;
        fe_mul(F, &ta, &alpha, &wbe);
        fe_mul(F, &tb, &beta, &wce);
        fe_add(F, &claim, &ta, &tb);                                 /* :233 */
        goff += gate_counts[L];
        coff += rounds * 3;
    }
    result = 1;
done:
    free(prev); free(w0);
    orc_transcript_free(t);
    return result;
}
