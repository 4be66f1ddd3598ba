/* gkr_wide.c -- TEST INFRASTRUCTURE (the checker; never linked or called by the product).
 *
 * gkr_protocol::prove (gkr/src/gkr_protocol.rs:26-143) restated for layers of ANY width, in time linear in the number of gates, straight
 * from the definition of the layer polynomial
 *
 *     f(b, c) = add~(b, c) (W~(b) + W~(c)) + mul~(b, c) W~(b) W~(c)                                  (utils.rs:8-21)
 *
 * with add~ / mul~ the multilinear extensions of the 0/1 wiring predicates (arithmetic_circuit.rs:126-163) folded over the output
 * variable(s) by the previous challenges (layer 0: gkr_protocol.rs:60-72; later layers alpha / beta combined: utils.rs:23-68).
 * A multilinear extension of a 0/1 table is the sum over its ones of the eq polynomial of their index, so for the round that binds variable
 * j of x = (b, c), with r_0 .. r_{j-1} already bound (sumcheck_gkr_protocol.rs:37-60),
 *
 *     g_j(X) = sum over gates g of   w_g * prod_{i<j} eq1(r_i, x_i(g)) * eq1(X, x_j(g)) * [ op(g) = add ?  Wb(X) + Wc(X)  :  Wb(X) Wc(X) ]
 *
 * where x(g) = left(g) || right(g) MSB first, w_g = alpha eq(rb', out(g)) + beta eq(rc', out(g)) written out as the product over the output
 * index's bits, and Wb(X) / Wc(X) are the values of W~ at the gate's left / right index with the bound prefix replaced by the challenges --
 * read from W folded by those challenges with the reference's own partial_evaluate (evaluation_form.rs:61-106), nothing else.  The round
 * polynomial is evaluated at X = 0, 1, 2 gate by gate (generate_round_univariate :113-143 evaluates at the same points), interpolated by the
 * reference's Lagrange routine (:46-50) and sent as coefficients (:52).
 *
 * What this deliberately does NOT share with the product's sparse prover (csrc/zkmle_gkr_sparse.hip): no eq TABLES (every weight is a
 * product of 1-variable factors per gate), no per-index H0 / H1 / A / M / C tables, no two-phase restatement of the polynomial, no nodes
 * 0 / 1 / infinity, no derived e(1).  On shapes small enough for the dense model (oracle/pymodel.py gkr_prove_wide) the two oracles must agree
 * (tests/test_oracle_gkr_wide.py); the product is compared with this one where the dense model cannot go (2^10 .. 2^18 gates per layer).
 */
#include "zk_internal.h"

#define EL(base, i) ((base) + (size_t)(i) * (size_t)F->n)

static void append_be(const field_t *F, orc_transcript *t, const fe *x) {
    uint8_t b[8 * MAXL];
    fe_to_be_bytes(F, b, x);
    orc_transcript_append(t, b, 8 * (size_t)F->n);
}
static void challenge(const field_t *F, orc_transcript *t, fe *out) {
    uint8_t d[32];
    orc_transcript_sample(t, d);                                     /* fiat_shamir_transcript.rs:29-33 */
    fe_from_le_bytes(F, out, d, 32);                                 /* :38-43 */
}
/* eq1(r, bit) = bit ? r : 1 - r */
static void eq1(const field_t *F, fe *o, const fe *r, unsigned bit, const fe *one) {
    if (bit) *o = *r;
    else fe_sub(F, o, one, r);
}
/* eq(r[0 .. k), idx) with variable 0 = the most significant of the k index bits (evaluation_form.rs:76-80) */
static void eq_index(const field_t *F, fe *o, const fe *r, size_t k, uint64_t idx, const fe *one) {
    *o = *one;
    for (size_t i = 0; i < k; i++) {
        fe f;
        eq1(F, &f, &r[i], (unsigned)((idx >> (k - 1 - i)) & 1), one);
        fe_mul(F, o, o, &f);
    }
}
static size_t ilog2z(size_t v) {
    size_t k = 0;
    while (((size_t)1 << k) < v) k++;
    return k;
}

int orc_gkr_prove_wide(int field, const orc_gate *gates, const size_t *gate_counts, size_t nlayers, const uint32_t *out_bits,
                       const uint64_t *inputs, size_t ninputs, uint64_t *circuit_output, uint64_t *claimed_sum,
                       uint64_t *layer_claims, uint64_t *coeffs, uint64_t *challenges, uint64_t *wb_evals, uint64_t *wc_evals,
                       uint64_t *output_challenges) {
    const field_t *F = orc_fld(field);
    if (!F) return ORC_E_ARG;
    if (nlayers == 0 || ninputs == 0 || (ninputs & (ninputs - 1)) || out_bits[0] == 0) return ORC_E_ARG;
    const size_t es = 8 * (size_t)F->n;
    fe one, two, minus_one;
    fe_one(F, &one);
    fe_add(F, &two, &one, &one);
    fe_neg(F, &minus_one, &one);
    int rc = ORC_OK;
    /* ---- Circuit::evaluate (arithmetic_circuit.rs:65-109) with the widths given: evs[l] = the wires layer l produces, evs[nlayers] = inputs */
    uint64_t **evs = (uint64_t **)calloc(nlayers + 1, sizeof *evs);
    size_t *width = (size_t *)calloc(nlayers + 1, sizeof *width);
    for (size_t l = 0; l < nlayers; l++) width[l] = (size_t)1 << out_bits[l];
    width[nlayers] = ninputs;
    evs[nlayers] = (uint64_t *)malloc(es * ninputs);
    memcpy(evs[nlayers], inputs, es * ninputs);
    {
        size_t goff = 0;
        for (size_t l = 0; l < nlayers; l++) goff += gate_counts[l];
        for (size_t l = nlayers; l-- > 0;) {                         /* :72 from the inputs up */
            goff -= gate_counts[l];
            evs[l] = (uint64_t *)calloc(width[l] * (size_t)F->n, 8);
            for (size_t g = 0; g < gate_counts[l]; g++) {
                const orc_gate *G = &gates[goff + g];
                if (G->left >= width[l + 1] || G->right >= width[l + 1] || G->out >= width[l]) { rc = ORC_E_ARG; goto done; }
                fe a, b, v, cur;
                fe_load(F, &a, EL(evs[l + 1], G->left));
                fe_load(F, &b, EL(evs[l + 1], G->right));
                if (G->op == 0) fe_add(F, &v, &a, &b); else fe_mul(F, &v, &a, &b);   /* :90-95 */
                fe_load(F, &cur, EL(evs[l], G->out));
                fe_add(F, &cur, &cur, &v);                           /* += :96 */
                fe_store(F, EL(evs[l], G->out), &cur);
            }
        }
    }
    memcpy(circuit_output, evs[0], es * width[0]);
    orc_transcript *t = orc_transcript_new();
    fe claim, alpha, beta;
    fe_zero(&alpha); fe_zero(&beta);
    const size_t k0 = out_bits[0];
    fe *ra = (fe *)malloc(sizeof(fe) * k0), *rb = NULL, *rcv = NULL;
    {
        uint8_t *bytes = (uint8_t *)malloc(es * width[0]);
        orc_mle_to_bytes(field, evs[0], width[0], bytes);
        orc_transcript_append(t, bytes, es * width[0]);              /* gkr_protocol.rs:49 */
        free(bytes);
        for (size_t i = 0; i < k0; i++) {                            /* :50, one challenge per output variable */
            challenge(F, t, &ra[i]);
            fe_store(F, EL(output_challenges, i), &ra[i]);
        }
        uint64_t o[MAXL];
        rc = orc_mle_evaluate(field, evs[0], width[0], output_challenges, k0, o);   /* :51 */
        if (rc != ORC_OK) goto done_t;
        fe_load(F, &claim, o);
    }
    size_t goff = 0, coff = 0, choff = 0, kprev = 0;
    for (size_t l = 0; l < nlayers; l++) {                           /* :57 */
        const size_t ng = gate_counts[l], ka = out_bits[l], k = ilog2z(width[l + 1]), rounds = 2 * k;
        const orc_gate *G = gates + goff;
        const uint64_t *W = evs[l + 1];
        if (l > 0 && kprev != ka) { rc = ORC_E_ARG; goto done_t; }  /* the previous layer's b / c challenges bind this layer's output variables */
        /* the gate's weight: the wiring predicates' output variables bound by the previous challenges, term by term */
        fe *e = (fe *)malloc(sizeof(fe) * (ng ? ng : 1));
#pragma omp parallel for schedule(static)
        for (size_t g = 0; g < ng; g++) {
            if (l == 0) eq_index(F, &e[g], ra, ka, G[g].out, &one);  /* :60-72 */
            else {                                                   /* utils.rs:23-68: alpha add(rb', ..) + beta add(rc', ..) */
                fe x, y;
                eq_index(F, &x, rb, kprev, G[g].out, &one);
                eq_index(F, &y, rcv, kprev, G[g].out, &one);
                fe_mul(F, &x, &x, &alpha);
                fe_mul(F, &y, &y, &beta);
                fe_add(F, &e[g], &x, &y);
            }
        }
        fe_store(F, EL(layer_claims, l), &claim);
        append_be(F, t, &claim);                                     /* sumcheck_gkr_protocol.rs:35 */
        /* W folded by the challenges bound so far: over b in rounds 0 .. k-1, over c in rounds k .. 2k-1 (evaluation_form.rs:61-106) */
        size_t blen = width[l + 1], clen = width[l + 1];
        uint64_t *Wb = (uint64_t *)malloc(es * blen), *Wc = (uint64_t *)malloc(es * clen);
        memcpy(Wb, W, es * blen);
        memcpy(Wc, W, es * clen);
        fe *r = (fe *)malloc(sizeof(fe) * (rounds ? rounds : 1));
        for (size_t j = 0; j < rounds; j++) {                        /* :37 */
            const int second = j >= k;
            const size_t jj = second ? j - k : j, shift = k - 1 - jj, half = (size_t)1 << shift;
            fe ev[3];
            fe_zero(&ev[0]); fe_zero(&ev[1]); fe_zero(&ev[2]);
            fe u;
            if (second) fe_load(F, &u, EL(Wb, 0));                   /* W~ at the bound b */
#pragma omp parallel
            {
                fe loc[3];
                fe_zero(&loc[0]); fe_zero(&loc[1]); fe_zero(&loc[2]);
#pragma omp for schedule(static) nowait
                for (size_t g = 0; g < ng; g++) {
                    const uint64_t idx = second ? G[g].right : G[g].left;
                    const unsigned bit = (unsigned)((idx >> shift) & 1);
                    const size_t low = (size_t)(idx & (half - 1));
                    const uint64_t *tab = second ? Wc : Wb;
                    fe v0, v1, v2, d, other;
                    fe_load(F, &v0, EL(tab, low));
                    fe_load(F, &v1, EL(tab, low + half));
                    fe_sub(F, &d, &v1, &v0);
                    fe_add(F, &v2, &v1, &d);                         /* the multilinear value at X = 2 */
                    if (second) other = u; else fe_load(F, &other, EL(W, G[g].right));
                    const fe *vx[3] = {&v0, &v1, &v2};
                    for (int X = 0; X < 3; X++) {                    /* generate_round_univariate :127-140: the points 0, 1, 2 */
                        fe q, body, term;                            /* eq1(X, bit): 1 - X or X */
                        if (X == 0) { if (bit) continue; q = one; }
                        else if (X == 1) { if (!bit) continue; q = one; }
                        else q = bit ? two : minus_one;
                        if (G[g].op == 0) fe_add(F, &body, vx[X], &other); else fe_mul(F, &body, vx[X], &other);
                        fe_mul(F, &term, &e[g], &body);
                        fe_mul(F, &term, &term, &q);
                        fe_add(F, &loc[X], &loc[X], &term);
                    }
                }
#pragma omp critical
                for (int X = 0; X < 3; X++) fe_add(F, &ev[X], &ev[X], &loc[X]);
            }
            uint64_t xs[3 * MAXL], ys[3 * MAXL], co[3 * MAXL];
            for (int X = 0; X < 3; X++) {
                fe xv;
                fe_from_u64(F, &xv, (uint64_t)X);                    /* :46-48 */
                fe_store(F, EL(xs, X), &xv);
                fe_store(F, EL(ys, X), &ev[X]);
            }
            rc = orc_uni_lagrange_interpolate(field, xs, ys, 3, co); /* :49-50 */
            if (rc != ORC_OK) { free(e); free(Wb); free(Wc); free(r); goto done_t; }
            uint8_t bytes[3 * 8 * MAXL];
            for (int c = 0; c < 3; c++) {
                fe cf;
                fe_load(F, &cf, EL(co, c));
                fe_to_le_bytes(F, bytes + (size_t)c * es, &cf);      /* univariate_to_bytes :145-150 */
                fe_store(F, EL(coeffs, coff + 3 * j + (size_t)c), &cf);
            }
            orc_transcript_append(t, bytes, 3 * es);                 /* :52 */
            challenge(F, t, &r[j]);                                  /* :55 */
            fe_store(F, EL(challenges, choff + j), &r[j]);
            /* :57: bind the variable -- every gate's weight takes the factor of its own bit, the table of this half is folded */
#pragma omp parallel for schedule(static)
            for (size_t g = 0; g < ng; g++) {
                const uint64_t idx = second ? G[g].right : G[g].left;
                fe f;
                eq1(F, &f, &r[j], (unsigned)((idx >> shift) & 1), &one);
                fe_mul(F, &e[g], &e[g], &f);
            }
            uint64_t **tabp = second ? &Wc : &Wb;
            size_t *lenp = second ? &clen : &blen;
            uint64_t *nx = (uint64_t *)malloc(es * (*lenp / 2 ? *lenp / 2 : 1));
            rc = mle_partial_evaluate(F, *tabp, *lenp, 0, &r[j], nx);
            free(*tabp);
            *tabp = nx;
            *lenp /= 2;
            if (rc != ORC_OK) { free(e); free(Wb); free(Wc); free(r); goto done_t; }
        }
        free(e); free(Wb); free(Wc);
        if (l < nlayers - 1) {                                       /* gkr_protocol.rs:109-133 */
            uint64_t o[MAXL];
            fe wbe, wce, ta, tb;
            rc = orc_mle_evaluate(field, W, width[l + 1], EL(challenges, choff), k, o);        /* utils.rs:78 */
            if (rc != ORC_OK) { free(r); goto done_t; }
            fe_load(F, &wbe, o);
            rc = orc_mle_evaluate(field, W, width[l + 1], EL(challenges, choff + k), k, o);    /* :79 */
            if (rc != ORC_OK) { free(r); goto done_t; }
            fe_load(F, &wce, o);
            fe_store(F, EL(wb_evals, l), &wbe);
            fe_store(F, EL(wc_evals, l), &wce);
            free(rb); free(rcv);
            rb = (fe *)malloc(sizeof(fe) * (k ? k : 1));
            rcv = (fe *)malloc(sizeof(fe) * (k ? k : 1));
            for (size_t i = 0; i < k; i++) { rb[i] = r[i]; rcv[i] = r[k + i]; }               /* :120-123 */
            kprev = k;
            append_be(F, t, &wbe);                                   /* :125 */
            challenge(F, t, &alpha);
            append_be(F, t, &wce);                                   /* :128 */
            challenge(F, t, &beta);
            fe_mul(F, &ta, &alpha, &wbe);
            fe_mul(F, &tb, &beta, &wce);
            fe_add(F, &claim, &ta, &tb);                             /* :132 */
        }
        free(r);
        goff += ng;
        coff += 3 * rounds;
        choff += rounds;
    }
    fe_store(F, claimed_sum, &claim);
done_t:
    free(ra); free(rb); free(rcv);
    orc_transcript_free(t);
done:
    for (size_t l = 0; l <= nlayers; l++) free(evs[l]);
    free(evs); free(width);
    return rc;
}
