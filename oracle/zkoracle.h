/*
 * zkoracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the multilinear hot path of
 * casweeney/zk-cryptography-research-implementations, following the reference
 * loop for loop.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product library
 * (libzkmle_amd.so) never links, loads or calls it.
 *
 * Parity status: the reference is Rust and cannot be built in this image (no
 * cargo/rustc), and all of its field / curve / hash arithmetic lives in
 * un-vendored crates (ark-ff 0.5.0, ark-ec 0.5.0, ark-bls12-381 0.5.0,
 * ark-bn254 0.5.0, sha3 0.10.8).  The oracle is therefore pinned by
 *   (1) every known-answer value the reference's own #[test]s hold for this
 *       path (tests/golden/reference_kats.json, cited file:line), and
 *   (2) public-spec KATs (Keccak-256) and algebraic identities.
 * Transcript bytes, challenges and G1 coordinates are NOT pinned by any
 * reference test ("parity unpinned" for those rows).
 *
 * Conventions:
 *   - a field element is `limbs` little-endian u64 limbs in Montgomery form,
 *     R = 2^(64*limbs)  (the arkworks Fp in-memory layout);
 *   - tables are arrays of such elements, index bit (n-1-v) <-> variable v.
 */
#ifndef ZKORACLE_H
#define ZKORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_FR381 = 0, ORC_FQ381 = 1, ORC_BN254_FQ = 2, ORC_BN254_FR = 3, ORC_NFIELDS = 4 };

/* status codes: 0 ok; negative = the reference would panic at this point */
enum {
    ORC_OK = 0,
    ORC_E_NOT_POW2 = -1,      /* evaluation_form.rs:13 */
    ORC_E_LEN_MISMATCH = -2,  /* evaluation_form.rs:112,129,149 */
    ORC_E_NVARS = -3,         /* product_polynomial.rs:16, sum_polynomial.rs:17 */
    ORC_E_NEED_TWO = -4,      /* product_polynomial.rs:59, sum_polynomial.rs:58 */
    ORC_E_KZG_LEN = -5,       /* multilinear_kzg.rs:29,55,60 */
    ORC_E_RANGE = -6,         /* implicit index panic */
    ORC_E_ARG = -7,
    ORC_E_NOMEM = -8
};

int orc_field_limbs(int field);
/* constants, canonical little-endian limbs: modulus, R mod m, R^2 mod m; inv = -m^-1 mod 2^64 */
int orc_field_constants(int field, uint64_t *modulus, uint64_t *r, uint64_t *r2, uint64_t *inv);

/* --- field elements (Montgomery limbs in / out) -------------------------- */
int orc_fe_from_u64(int field, uint64_t v, uint64_t *out);
/* F::from_le_bytes_mod_order [ark-ff]; any length */
int orc_fe_from_le_bytes_mod_order(int field, const uint8_t *bytes, size_t n, uint64_t *out);
/* into_bigint().to_bytes_be() / to_bytes_le(): 8*limbs bytes */
int orc_fe_to_bytes_be(int field, const uint64_t *a, uint8_t *out);
int orc_fe_to_bytes_le(int field, const uint64_t *a, uint8_t *out);
int orc_fe_add(int field, const uint64_t *a, const uint64_t *b, uint64_t *out);
int orc_fe_sub(int field, const uint64_t *a, const uint64_t *b, uint64_t *out);
int orc_fe_mul(int field, const uint64_t *a, const uint64_t *b, uint64_t *out);
int orc_fe_neg(int field, const uint64_t *a, uint64_t *out);
int orc_fe_inv(int field, const uint64_t *a, uint64_t *out);
/* vectorised helpers for fixtures: n elements */
int orc_vec_from_canonical(int field, const uint64_t *canon, size_t n, uint64_t *mont);
int orc_vec_to_canonical(int field, const uint64_t *mont, size_t n, uint64_t *canon);

/* --- Keccak-256 and the Fiat-Shamir transcript --------------------------- */
void orc_keccak256(const uint8_t *data, size_t n, uint8_t out[32]);
typedef struct orc_transcript orc_transcript;
orc_transcript *orc_transcript_new(void);                    /* fiat_shamir_transcript.rs:12 */
void orc_transcript_free(orc_transcript *t);
void orc_transcript_append(orc_transcript *t, const uint8_t *data, size_t n);   /* :22 */
void orc_transcript_sample(orc_transcript *t, uint8_t out[32]);                 /* :29 */
int orc_transcript_challenge(orc_transcript *t, int field, uint64_t *out);      /* :38 */

/* --- MultilinearPolynomial (polynomials/src/multilinear/evaluation_form.rs) */
int orc_mle_new_check(size_t len);                                                    /* :12 */
int orc_mle_partial_evaluate(int field, const uint64_t *poly, size_t len, size_t var,
                             const uint64_t *value, uint64_t *out /* len/2 */);       /* :61 */
int orc_mle_evaluate(int field, const uint64_t *poly, size_t len, const uint64_t *values,
                     size_t nvalues, uint64_t *out /* 1 */);                          /* :21 */
int orc_mle_to_bytes(int field, const uint64_t *poly, size_t len, uint8_t *out);      /* :35 */
int orc_mle_scalar_mul(int field, const uint64_t *poly, size_t len, const uint64_t *s,
                       uint64_t *out);                                                /* :49 */
int orc_mle_add(int field, const uint64_t *a, size_t la, const uint64_t *b, size_t lb,
                uint64_t *out);                                                       /* :145 */
int orc_mle_tensor_add(int field, const uint64_t *wb, size_t lb, const uint64_t *wc, size_t lc,
                       uint64_t *out /* lb*lc */);                                    /* :108 */
int orc_mle_tensor_mul(int field, const uint64_t *wb, size_t lb, const uint64_t *wc, size_t lc,
                       uint64_t *out);                                                /* :125 */
int orc_vec_sum(int field, const uint64_t *a, size_t n, uint64_t *out);

/* --- univariate (polynomials/src/univariate/dense_univariate.rs) ---------- */
int orc_uni_evaluate(int field, const uint64_t *coeffs, size_t n, const uint64_t *x,
                     uint64_t *out);                                                  /* :57 */
/* lagrange_interpolate :74 ; out has n coefficients */
int orc_uni_lagrange_interpolate(int field, const uint64_t *xs, const uint64_t *ys, size_t n,
                                 uint64_t *out);

/* --- basic sumcheck (sumcheck_protocol/src/basic_sumcheck) ---------------- */
/* Prover::init + prove (prover.rs:22,35).  round_polys: nvars*2 elements;
 * challenges (nvars, diagnostic: not part of the reference's proof) may be NULL */
int orc_sumcheck_basic_prove(int field, const uint64_t *table, size_t len, uint64_t *claimed_sum,
                             uint64_t *round_polys, uint64_t *challenges);
/* Verifier::verify (verifier.rs:23): returns 1 / 0, negative on panic */
int orc_sumcheck_basic_verify(int field, const uint64_t *table, size_t len,
                              const uint64_t *claimed_sum, const uint64_t *round_polys,
                              size_t nrounds);
/* split_polynomial_and_sum_each prover.rs:74 */
int orc_split_and_sum(int field, const uint64_t *table, size_t len, uint64_t *out2);

/* --- composed polynomials + GKR sumcheck ---------------------------------- */
/* A SumPolynomial is nprod ProductPolynomials of nfac MLEs each, all of length len,
 * passed as one array tables[(p*nfac+f)*len + i]. */
int orc_sumpoly_evaluate(int field, const uint64_t *tables, size_t nprod, size_t nfac, size_t len,
                         const uint64_t *values, size_t nvalues, uint64_t *out);
int orc_sumpoly_reduce(int field, const uint64_t *tables, size_t nprod, size_t nfac, size_t len,
                       uint64_t *out /* len */);   /* add_polynomials_element_wise sum_polynomial.rs:57 */
/* generate_round_univariate sumcheck_gkr_protocol.rs:113 ; out: nfac+1 evaluations */
int orc_gkr_round_univariate(int field, const uint64_t *tables, size_t nprod, size_t nfac,
                             size_t len, uint64_t *out);
/* prove sumcheck_gkr_protocol.rs:24 ; round_coeffs: nvars*(nfac+1), challenges: nvars */
int orc_sumcheck_gkr_prove(int field, const uint64_t *tables, size_t nprod, size_t nfac, size_t len,
                           const uint64_t *claimed_sum, orc_transcript *t, uint64_t *round_coeffs,
                           uint64_t *challenges);
/* verify :69 ; returns 1/0; writes challenges (nrounds) and last_claimed_sum */
int orc_sumcheck_gkr_verify(int field, const uint64_t *claimed_sum, const uint64_t *round_coeffs,
                            size_t nrounds, size_t ncoef, orc_transcript *t, uint64_t *challenges,
                            uint64_t *last_claimed_sum);

/* --- circuit + GKR (circuit/src/arithmetic_circuit.rs, gkr/src) ----------- */
typedef struct { uint64_t left, right, out, op; /* 0 = Add, 1 = Mul */ } orc_gate;
size_t orc_num_layer_variables(size_t layer_index);                          /* :166 */
size_t orc_wiring_index(size_t layer_index, size_t a, size_t b, size_t c);   /* :180 */
/* Circuit::evaluate :65.  layer_sizes[nlayers+1] receives the length of each layer evaluation
 * (index 0 = output ... nlayers = inputs); evals receives them concatenated (caller sizes it
 * with orc_circuit_eval_size). */
size_t orc_circuit_eval_size(const orc_gate *gates, const size_t *gate_counts, size_t nlayers,
                             size_t ninputs);
int orc_circuit_evaluate(int field, const orc_gate *gates, const size_t *gate_counts,
                         size_t nlayers, const uint64_t *inputs, size_t ninputs,
                         size_t *layer_sizes, uint64_t *evals);
/* add_i_and_mul_i_mle :126 ; each out has 2^orc_num_layer_variables(layer) elements */
int orc_circuit_add_mul_mle(int field, const orc_gate *layer_gates, size_t ngates,
                            size_t layer_index, uint64_t *add_i, uint64_t *mul_i);

/* gkr_protocol::prove gkr_protocol.rs:26.  Flattened proof (Proof :17-23):
 *   circuit_output[*output_len], claimed_sum[1] (the FINAL claimed_sum field),
 *   layer_claims[nlayers]  = SumcheckProverProof.claimed_sum per layer,
 *   coeffs  = per layer rounds(L)*3 coefficients, concatenated (rounds(L) = 2*(L+1)),
 *   challenges = per layer rounds(L) elements, concatenated,
 *   wb_evals / wc_evals [nlayers-1]. */
size_t orc_gkr_rounds(size_t layer_index);
int orc_gkr_prove(int field, const orc_gate *gates, const size_t *gate_counts, size_t nlayers,
                  const uint64_t *inputs, size_t ninputs, uint64_t *circuit_output,
                  size_t *output_len, uint64_t *claimed_sum, uint64_t *layer_claims,
                  uint64_t *coeffs, uint64_t *challenges, uint64_t *wb_evals, uint64_t *wc_evals);
/* gkr_protocol::prove for layers of any width, linear in the number of gates, straight from the definition of the layer polynomial
 * (gkr_wide.c): layer l has 2^out_bits[l] outputs and reads the 2^out_bits[l+1] wires of the next layer (the ninputs inputs for the last);
 * out_bits[0] >= 1 output challenges; rounds(l) = 2 * log2(width of layer l + 1).  Same flattened proof as orc_gkr_prove. */
int orc_gkr_prove_wide(int field, const orc_gate *gates, const size_t *gate_counts, size_t nlayers, const uint32_t *out_bits,
                       const uint64_t *inputs, size_t ninputs, uint64_t *circuit_output, uint64_t *claimed_sum,
                       uint64_t *layer_claims, uint64_t *coeffs, uint64_t *challenges, uint64_t *wb_evals, uint64_t *wc_evals,
                       uint64_t *output_challenges);
/* gkr_protocol::verify :146 ; 1/0 */
int orc_gkr_verify(int field, const orc_gate *gates, const size_t *gate_counts, size_t nlayers,
                   const uint64_t *inputs, size_t ninputs, const uint64_t *circuit_output,
                   size_t output_len, const uint64_t *layer_claims, const uint64_t *coeffs,
                   const uint64_t *challenges, const uint64_t *wb_evals, const uint64_t *wc_evals);

/* --- BLS12-381 G1 + multilinear KZG (multilinear_kzg/src) ------------------
 * Affine point = 12 u64: x[6] | y[6] in Fq Montgomery form; infinity = all zero
 * (x = y = 0 is not on y^2 = x^3 + 4).  Group results are compared as affine. */
int orc_g1_generator(uint64_t *out12);
int orc_g1_is_on_curve(const uint64_t *p12);
int orc_g1_add(const uint64_t *p12, const uint64_t *q12, uint64_t *out12);
int orc_g1_neg(const uint64_t *p12, uint64_t *out12);
/* PrimeGroup::mul_bigint with an Fr scalar given in Montgomery form */
int orc_g1_mul_fr(const uint64_t *p12, const uint64_t *scalar_fr, uint64_t *out12);
/* compute_lagrange_basis trusted_setup.rs:24 ; out 2^ntaus Fr elements */
int orc_kzg_lagrange_basis(const uint64_t *taus, size_t ntaus, uint64_t *out);
/* compute_g1_powers_of_tau trusted_setup.rs:51 ; out 2^ntaus affine points */
int orc_kzg_setup_g1(const uint64_t *taus, size_t ntaus, uint64_t *out_points);
/* commit_to_polynomial multilinear_kzg.rs:25 (naive sum of mul_bigint) */
int orc_kzg_commit(const uint64_t *values, size_t len, const uint64_t *g1_points, size_t npoints,
                   uint64_t *out12);
/* open_and_prove :50 (naive, blown-up quotients). proofs: nvars affine points */
int orc_kzg_open(const uint64_t *values, size_t len, const uint64_t *g1_points, size_t npoints,
                 const uint64_t *opening, size_t nopen, size_t n_g2, uint64_t *evaluation,
                 uint64_t *proofs);
/* quotient evaluations Q_i as tables (diagnostic for the algebraic identity check):
 * round i writes 2^(n-1-i) elements at out + offset_i, offsets cumulative */
int orc_kzg_quotients(const uint64_t *values, size_t len, const uint64_t *opening, size_t nopen,
                      uint64_t *out);

/* --- CPU baseline timing helpers (bench.py cpu_baseline leg) --------------- */
/* fold the table `reps` times with the reference's allocation pattern; returns seconds */
double orc_bench_fold(int field, const uint64_t *table, size_t len, const uint64_t *r, int reps);
double orc_bench_fold_mt(int field, const uint64_t *table, size_t len, const uint64_t *r, int reps, int *threads_used);
double orc_bench_commit_naive(const uint64_t *values, size_t len, const uint64_t *g1_points);
/* best-effort CPU MSM (bucket method, OpenMP over (window, slice) items): same group element as orc_kzg_commit.  Not reference
 * code (the reference has no MSM routine): the "all cores" CPU baseline of BASELINE.md section 3.2 */
int orc_msm_pippenger(const uint64_t *values, size_t len, const uint64_t *g1_points, int window_bits, int slices, uint64_t *out12,
                      int *threads_used);
double orc_bench_pippenger_mt(const uint64_t *values, size_t len, const uint64_t *g1_points, int window_bits, int slices,
                              uint64_t *out12, int *threads_used);

#ifdef __cplusplus
}
#endif
#endif
