/*
 * zkmle.h -- C ABI of libzkmle_amd.so: the MI355X (gfx950) implementation of the multilinear
 * hot path of casweeney/zk-cryptography-research-implementations.
 *
 * The reference has no FFI layer: its path is reached through generic Rust items.
 * Each entry point below replaces the reference item cited next to it; the Rust shim that binds
 * them is shown in INTEGRATION.md (rust_shim/ holds its source).
 *
 * Conventions
 *   - field elements: `limbs` little-endian u64 limbs in Montgomery form, R = 2^(64*limbs) -- the
 *     in-memory layout of an arkworks `Fp`, so `&[Fr]` / `Vec<F>` can be passed by pointer;
 *   - tables: contiguous arrays of elements; index bit (n-1-v) <-> variable v (variable 0 = MSB),
 *     evaluation_form.rs:76-80;
 *   - every function returns a zk_status; precondition failures that PANIC in the reference
 *     return the matching negative code and never abort (zk_status_message gives the
 *     reference's panic text);
 *   - `zk_table` handles own device (HBM) memory; host pointers stay owned by the caller;
 *   - a handle is used by one thread at a time; the library is re-entrant after zk_init.
 * The library NEVER falls back to a CPU path: without a usable HIP device every compute entry
 * point returns ZK_E_NO_DEVICE.
 */
#ifndef ZKMLE_H
#define ZKMLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { ZK_FR381 = 0, ZK_FQ381 = 1, ZK_BN254_FQ = 2, ZK_BN254_FR = 3 } zk_field;

typedef enum {
    ZK_OK = 0,
    ZK_E_NOT_POW2 = -1,     /* "Evaluated values must be a power of 2"  evaluation_form.rs:13 */
    ZK_E_LEN_MISMATCH = -2, /* evaluation_form.rs:112,129-132,149-153 */
    ZK_E_NVARS = -3,        /* "different number of variables" product_polynomial.rs:16-21, sum_polynomial.rs:17-23 */
    ZK_E_NEED_TWO = -4,     /* product_polynomial.rs:59-62, sum_polynomial.rs:58-61 */
    ZK_E_KZG_LEN = -5,      /* multilinear_kzg.rs:29-33,55-64 */
    ZK_E_RANGE = -6,        /* implicit index / underflow panics */
    ZK_E_ARG = -7,          /* bad argument at the C boundary (no reference counterpart) */
    ZK_E_NOMEM = -8,
    ZK_E_NO_DEVICE = -9,    /* no usable HIP device: the product path has no CPU fallback */
    ZK_E_HIP = -10,         /* a HIP runtime call failed; see zk_last_error */
    ZK_E_NOT_INIT = -11,    /* "Can't prove without init" prover.rs:36 */
    ZK_E_COMM = -12         /* an RCCL call or a caller-supplied exchange callback failed; see zk_last_error */
} zk_status;

const char *zk_status_message(int status);
const char *zk_last_error(void);      /* thread-local detail for ZK_E_HIP */
const char *zk_version(void);

/* ---- device ------------------------------------------------------------------------------- */
int zk_device_count(int *count);
int zk_init(int device);              /* select the device for the calling thread's later calls */
int zk_field_limbs(int field);        /* u64 limbs per element (4 or 6), negative on bad field */
int zk_device_synchronize(void);
/* Streams and threads: every launch, copy and wait of the library runs on the CALLING THREAD's current stream -- the
 * default (NULL) stream unless zk_set_stream was called on that thread.  A handle is used by one thread at a time; threads
 * that set different streams run their calls concurrently on the device (scratch buffers, staging buffers and the *_stats
 * are per thread; cached scratch blocks are reused only on the stream they last served).  The `stream` argument of the
 * zk_mle_* calls, when not NULL, overrides the thread's stream for that call. */
int zk_set_stream(void *hip_stream);
void *zk_get_stream(void);
int zk_release_cached_memory(void);   /* frees the library's cached per-call scratch (MSM workspaces) on the current device */

/* ---- device-resident tables ----------------------------------------------------------------
 * MultilinearPolynomial<F>{evaluated_values: Vec<F>}  evaluation_form.rs:7-18 */
typedef struct zk_table zk_table;
int zk_table_alloc(int field, size_t len, zk_table **out);                 /* uninitialised */
int zk_table_upload(int field, const uint64_t *host, size_t len, zk_table **out);  /* MultilinearPolynomial::new :12 (asserts pow2) */
/* plain vector upload without the power-of-two assert (MSM scalars, W layers): any len >= 1 */
int zk_table_upload_raw(int field, const uint64_t *host, size_t len, zk_table **out);
int zk_table_download(const zk_table *t, uint64_t *host);
int zk_table_free(zk_table *t);
size_t zk_table_len(const zk_table *t);
int zk_table_field(const zk_table *t);
void *zk_table_device_ptr(zk_table *t);                                    /* raw HBM pointer */
int zk_table_wrap(int field, void *device_ptr, size_t len, zk_table **out); /* non-owning view of caller's HBM */
int zk_table_clone(const zk_table *t, zk_table **out);
/* synthetic benchmark data: element i = SplitMix64-derived 4 (6) words reduced mod p */
int zk_table_fill_random(zk_table *t, uint64_t seed);
/* local entry j <- element (first + j * stride) of the stream `seed`: rank g of G generates its low-bit shard of a global
 * table with (first, stride) = (g, G), its contiguous slice of an MSM's scalars with (g * n / G, 1) */
int zk_table_fill_random_strided(zk_table *t, uint64_t seed, size_t first, size_t stride);
/* host mirror of the generator (same bytes), for parity tests */
int zk_host_fill_random(int field, uint64_t seed, size_t first, size_t count, uint64_t *out);

/* ---- MLE operations on device tables --------------------------------------------------------
 * every `stream` argument is a hipStream_t (NULL = the null stream) */
/* partial_evaluate  evaluation_form.rs:61-106.  out: len/2 elements (out != in) */
int zk_mle_fold(const zk_table *in, size_t var, const uint64_t *value, zk_table *out, void *stream);
/* the same on raw device pointers (HBM-resident slices of a caller-managed buffer) */
int zk_mle_fold_ptr(int field, const void *d_in, size_t len, size_t var, const uint64_t *value,
                    void *d_out, void *stream);
/* evaluate  evaluation_form.rs:21-33 : nvalues successive var-0 folds, element 0 */
int zk_mle_evaluate(const zk_table *t, const uint64_t *values, size_t nvalues, uint64_t *out);
/* convert_to_bytes  :35-43 : canonical big-endian bytes, len*8*limbs */
int zk_mle_to_bytes(const zk_table *t, uint8_t *host_out);
/* scalar_mul :49, add_polynomials :145, polynomial_tensor_add :108, polynomial_tensor_mul :125 */
int zk_mle_scalar_mul(const zk_table *a, const uint64_t *scalar, zk_table *out, void *stream);
int zk_mle_add(const zk_table *a, const zk_table *b, zk_table *out, void *stream);
int zk_mle_sub_scalar(const zk_table *a, const uint64_t *scalar, zk_table *out, void *stream); /* multilinear_kzg.rs:74-78 */
int zk_mle_tensor_add(const zk_table *wb, const zk_table *wc, zk_table *out, void *stream);
int zk_mle_tensor_mul(const zk_table *wb, const zk_table *wc, zk_table *out, void *stream);
/* iter().sum()  prover.rs:28 ; and split_polynomial_and_sum_each prover.rs:74-89 (out2: 2 elements) */
int zk_mle_sum(const zk_table *t, uint64_t *out);
int zk_mle_half_sums(const zk_table *t, uint64_t *out2);
/* fused sumcheck round: fold by `value` AND return the folded table's two half sums (the next
 * round's univariate) in one pass over HBM */
int zk_mle_fold_half_sums(const zk_table *in, const uint64_t *value, zk_table *out, uint64_t *out2,
                          void *stream);

/* ---- stateless host-buffer conveniences (upload, compute on the GPU, download) --------------- */
int zk_host_partial_evaluate(int field, const uint64_t *poly, size_t len, size_t var,
                             const uint64_t *value, uint64_t *out);
int zk_host_evaluate(int field, const uint64_t *poly, size_t len, const uint64_t *values,
                     size_t nvalues, uint64_t *out);

/* ---- host-side helpers (CPU, tiny: control path) -------------------------------------------- */
int zk_fe_from_u64(int field, uint64_t v, uint64_t *out);
int zk_fe_to_bytes_be(int field, const uint64_t *a, uint8_t *out);
int zk_fe_from_le_bytes_mod_order(int field, const uint8_t *bytes, size_t n, uint64_t *out);
int zk_fe_add(int field, const uint64_t *a, const uint64_t *b, uint64_t *out);
int zk_fe_sub(int field, const uint64_t *a, const uint64_t *b, uint64_t *out);
int zk_fe_mul(int field, const uint64_t *a, const uint64_t *b, uint64_t *out);
int zk_fe_inv(int field, const uint64_t *a, uint64_t *out);
int zk_vec_from_canonical(int field, const uint64_t *canon, size_t n, uint64_t *mont);
int zk_vec_to_canonical(int field, const uint64_t *mont, size_t n, uint64_t *canon);

/* ---- Fiat-Shamir transcript (host; transcripts/src/fiat_shamir/fiat_shamir_transcript.rs:5-43) ---- */
typedef struct zk_transcript zk_transcript;
int zk_transcript_new(zk_transcript **out);                                        /* Transcript::new :12 */
int zk_transcript_free(zk_transcript *t);
int zk_transcript_append(zk_transcript *t, const uint8_t *data, size_t n);         /* append :22 */
int zk_transcript_sample(zk_transcript *t, uint8_t out32[32]);                     /* sample_random_challenge :29 */
int zk_transcript_challenge(zk_transcript *t, int field, uint64_t *out);           /* random_challenge_as_field_element :38 */
int zk_keccak256(const uint8_t *data, size_t n, uint8_t out32[32]);
/* the running sponge as 25 Keccak lanes + the fill of the open block: what one rank hands to the others after absorbing
 * input only it has seen (the whole-table absorb of a sharded Prover::prove) */
int zk_transcript_export_state(const zk_transcript *t, uint64_t lanes25[25], uint32_t *fill);
int zk_transcript_import_state(zk_transcript *t, const uint64_t lanes25[25], uint32_t fill);

/* ---- univariate helpers (host; polynomials/src/univariate/dense_univariate.rs) ------------------ */
int zk_uni_evaluate(int field, const uint64_t *coeffs, size_t n, const uint64_t *x, uint64_t *out);        /* :57 */
int zk_uni_lagrange_interpolate(int field, const uint64_t *xs, const uint64_t *ys, size_t n, uint64_t *out); /* :74 */

/* ---- basic sumcheck (sumcheck_protocol/src/basic_sumcheck) ------------------------------------- */
/* Host-clock split of the last zk_sumcheck_basic_prove / zk_sumcheck_gkr_prove / zk_sumcheck_gkr_rounds call made on the
 * calling thread: `ms_absorb` = the whole-table transcript absorb (prover.rs:38-39, sequential Keccak on the host),
 * `ms_rounds` = every round (kernels + device-side transcript) up to the single synchronisation. */
typedef struct {
    uint32_t rounds;
    float ms_absorb, ms_rounds;
} zk_sumcheck_stats;
int zk_sumcheck_last_stats(zk_sumcheck_stats *out);
/* Fault injection for tests: the NEXT proof (of any thread) whose transcript steps run on the host finds its host side deaf for
 * `milliseconds` -- as if the process had been descheduled.  The kernels waiting for their challenge give up after their spin
 * budget (2-3 s), every kernel still ends, and the proving call returns ZK_E_HIP with zk_last_error naming the cause; the
 * thread's next proof works.  The reference has no analogue (its prover cannot stall: it is one thread).  Process-global, so it is
 * inert -- returns ZK_E_ARG -- unless the process runs with ZK_ENABLE_FAULT_INJECTION=1 in its environment. */
int zk_debug_stall_service_once(int milliseconds);

/* Prover::init + Prover::prove  prover.rs:22-71.  The table stays in HBM; per round one fused
 * fold + half-sums kernel; after the table absorb the transcript lives on the device (one host
 * synchronisation per proof, csrc/dev_transcript.cuh).  round_polys: nvars*2 elements
 * (SumcheckProof.round_univariate_polynomials); challenges (nvars) is diagnostic, may be NULL. */
int zk_sumcheck_basic_prove(const zk_table *table, uint64_t *claimed_sum, uint64_t *round_polys,
                            uint64_t *challenges);
/* the same on the CALLER's transcript -- `Prover { transcript, .. }` is a public field of the reference's prover (prover.rs:7-13) that prove()
 * appends to (:38-58): the transcript ends in the state the reference's ends in, whatever it had absorbed before */
int zk_sumcheck_basic_prove_on(const zk_table *table, zk_transcript *transcript, uint64_t *claimed_sum, uint64_t *round_polys,
                               uint64_t *challenges);
/* Verifier::verify  verifier.rs:23-71 (its final `evaluate` is the same GPU fold); *ok = 1 / 0 */
int zk_sumcheck_basic_verify(const zk_table *table, const uint64_t *claimed_sum,
                             const uint64_t *round_polys, size_t nrounds, int *ok);

/* ---- composed polynomials + GKR sumcheck ----------------------------------------------------------
 * A SumPolynomial (polynomials/src/composed/sum_polynomial.rs:7-9) of `nprod` ProductPolynomials
 * (product_polynomial.rs:6-8) with `nfac` MLEs each is passed as tables[p * nfac + f].
 * SumPolynomial::new / ProductPolynomial::new assert equal variable counts (ZK_E_NVARS). */
int zk_sumpoly_evaluate(const zk_table *const *tables, size_t nprod, size_t nfac,
                        const uint64_t *values, size_t nvalues, uint64_t *out);    /* sum_polynomial.rs:30 */
int zk_sumpoly_reduce(const zk_table *const *tables, size_t nprod, size_t nfac, zk_table *out); /* :57 add_polynomials_element_wise */
/* ProductPolynomial::multiply_polynomials_element_wise (product_polynomial.rs:58-73): out[i] = prod_f tables[f][i]; nfac >= 2
 * (the reference asserts "more than one polynomial required for mul operation") */
int zk_prodpoly_reduce(const zk_table *const *tables, size_t nfac, zk_table *out);
/* generate_round_univariate sumcheck_gkr_protocol.rs:113-143 ; out: nfac+1 evaluations at 0..nfac.
 * out == NULL (here and in zk_sumpoly_fold_round_evals' out_evals): the pass over the tables is only ENQUEUED on the current
 * stream -- no reduction of the per-workgroup partials, no read-back -- which is how bench.py times the round kernels alone. */
int zk_sumpoly_round_evals(const zk_table *const *tables, size_t nprod, size_t nfac, uint64_t *out);
/* one fused prover round on caller-managed tables: fold every table of `in` by `value` into `out`
 * (len/2 each) and return the NEXT round's nfac+1 evaluations of the folded tables (len >= 4).
 * Building block of the multi-GPU prover, where the host combines per-shard evaluations. */
int zk_sumpoly_fold_round_evals(const zk_table *const *in, zk_table *const *out, size_t nprod, size_t nfac,
                                const uint64_t *value, uint64_t *out_evals);
/* prove  sumcheck_gkr_protocol.rs:24-67.  round_coeffs: nvars*(nfac+1) coefficients, challenges: nvars.
 * The caller's tables are not modified (the reference clones, :33). */
int zk_sumcheck_gkr_prove(const zk_table *const *tables, size_t nprod, size_t nfac,
                          const uint64_t *claimed_sum, zk_transcript *t, uint64_t *round_coeffs,
                          uint64_t *challenges);
/* the rounds of `prove` without the leading claimed-sum append (:37-60), for provers that run one
 * sumcheck in several phases on one transcript (the sparse GKR prover).  final_values (may be NULL):
 * the nprod*nfac one-entry tables left after the last fold. */
int zk_sumcheck_gkr_rounds(const zk_table *const *tables, size_t nprod, size_t nfac, zk_transcript *t,
                           uint64_t *round_coeffs, uint64_t *challenges, uint64_t *final_values);
/* the same rounds where the SECOND factor of some two-factor products is a constant: tables[p * 2 + 1] == NULL means
 * "const_factors[p] at every index" (nfac must be 2, const_factors: nprod elements).  A product with a constant factor is a
 * linear term sum_i c X(i); its table is never materialised, streamed or folded, and the proof is the one the rounds above give
 * with that table filled with the constant (tests/test_gpu_sumcheck.py).  The sparse GKR prover's phases W H1 + H0 * 1 and
 * C W + A * u run through it with three tables instead of four. */
int zk_sumcheck_gkr_rounds_cf(const zk_table *const *tables, size_t nprod, size_t nfac, const uint64_t *const_factors,
                              zk_transcript *t, uint64_t *round_coeffs, uint64_t *challenges, uint64_t *final_values);
/* verify :69-105 (host only: O(rounds) field operations) */
int zk_sumcheck_gkr_verify(int field, const uint64_t *claimed_sum, const uint64_t *round_coeffs,
                           size_t nrounds, size_t ncoef, zk_transcript *t, uint64_t *challenges,
                           uint64_t *last_claimed_sum, int *ok);

/* ---- device-resident rounds as a handle: sharded (one process per GPU) sumcheck provers ------------------
 * each GPU owns the table entries i == rank (mod G); a round is local except for the sum of 2 (basic) or
 * d + 1 (GKR) evaluations over the ranks.  With this handle that sum is ONE all-reduce(SUM) of `zk_rounds_limbs_len`
 * 64-bit words in device memory (RCCL; each word holds a 32-bit limb of a lazy sum, so element-wise integer
 * addition is exact), and the transcript step runs on every rank's device (csrc/dev_transcript.cuh): no host round
 * trip per round.  Sequence: zk_rounds_evals -> all-reduce -> zk_rounds_absorb, then per round zk_rounds_fold_evals
 * -> all-reduce -> zk_rounds_absorb; when one entry per rank is left, gather the G entries on every rank,
 * zk_rounds_evals + zk_rounds_absorb on the gathered tables (no all-reduce: replicated) and zk_rounds_tail.
 * mode 0 = basic sumcheck (prover.rs:35-71; nprod = nfac = 1, messages = the two half sums, the claimed sum is
 * absorbed before round 0), mode 1 = GKR sumcheck rounds (sumcheck_gkr_protocol.rs:37-60).  `t` is read at creation
 * (everything absorbed so far) and written back by zk_rounds_collect; keep it alive and untouched in between (by default the
 * transcript step of a round runs on the calling thread's host side, on `t` itself: the kernels post the summed evaluations to a
 * pinned mailbox and wait for the challenge -- ZK_HOST_TRANSCRIPT=0 keeps the step on the device). */
typedef struct zk_rounds zk_rounds;
int zk_rounds_new(int field, int mode, size_t nprod, size_t nfac, size_t nrounds, zk_transcript *t, zk_rounds **out);
int zk_rounds_free(zk_rounds *r);
size_t zk_rounds_limbs_len(const zk_rounds *r);                     /* capacity in words: (nfac + 1) * (limbs32 + 1); mode 0: 16 * (limbs32 + 1) */
/* evaluations of the next round from the CURRENT tables (no fold) -> limbs_dev (device memory) */
int zk_rounds_evals(zk_rounds *r, const zk_table *const *tables, uint64_t *limbs_dev);
/* fold every table by the last absorbed round's challenge (device-resident) into `out`, and the next round's
 * evaluations -> limbs_dev; tables of 2 entries are only folded (limbs_dev untouched, may be NULL) */
int zk_rounds_fold_evals(zk_rounds *r, const zk_table *const *in, zk_table *const *out, uint64_t *limbs_dev);
/* transcript step of the next round on the (summed) limbs: message, absorb, challenge */
int zk_rounds_absorb(zk_rounds *r, const uint64_t *limbs_dev);
/* every remaining round in one launch on tables every rank holds in full (<= 2048 entries); the last absorbed
 * round's challenge folds first */
int zk_rounds_tail(zk_rounds *r, const zk_table *const *tables);
/* Basic sumcheck (mode 0) with the host-assisted transcript step: SEVERAL rounds per pass and per all-reduce
 * (csrc/basic_multi.cuh).  The m rounds after a pass are the basic sumcheck on the table's 2^m segment sums (folding the top
 * variable commutes with summing out the low ones), so: zk_rounds_multi_evals (2^m segment sums -> 2^m * (limbs32 + 1) words)
 * -> all-reduce -> zk_rounds_multi_absorb (m transcript steps, one exchange) -> zk_rounds_multi_fold_evals (fold the k = m
 * variables just absorbed, prover.rs:61-63 k times, and the 2^m_next segment sums of the output; m_next = 0: none) -> ...;
 * once the global table has <= 2048 entries, gather it on every rank and zk_rounds_multi_tail runs every round left (none of
 * them started) in one launch.  zk_rounds_multi_max = the largest m accepted, 0 when the handle cannot do this (mode 1, or the
 * transcript step on the device): use the one-round sequence above then.  Same messages, same bytes absorbed.
 * limbs_dev == NULL in zk_rounds_multi_evals / zk_rounds_multi_fold_evals (with m_next > 0) means ONE rank: nothing is all-reduced, the
 * pass's last workgroup runs the exchange itself and the rounds count as absorbed (no zk_rounds_multi_absorb for them). */
unsigned zk_rounds_multi_max(const zk_rounds *r);
int zk_rounds_multi_evals(zk_rounds *r, const zk_table *table, unsigned m, uint64_t *limbs_dev);
int zk_rounds_multi_absorb(zk_rounds *r, const uint64_t *limbs_dev, unsigned m);
int zk_rounds_multi_fold_evals(zk_rounds *r, const zk_table *in, zk_table *out, unsigned k, unsigned m_next, uint64_t *limbs_dev);
int zk_rounds_multi_tail(zk_rounds *r, const zk_table *table);
/* the single synchronisation: messages (nrounds x (nfac + 1) elements), challenges (nrounds), the claimed sum
 * (mode 0) and, after zk_rounds_tail, the nprod * nfac fully folded values; any pointer may be NULL */
int zk_rounds_collect(zk_rounds *r, zk_transcript *t, uint64_t *claimed_sum, uint64_t *messages, uint64_t *challenges,
                      uint64_t *final_values);

/* ---- layered circuit + GKR prover (circuit/src/arithmetic_circuit.rs, gkr/src/gkr_protocol.rs) ----
 * Gate :9-15 (op 0 = Add, 1 = Mul); a circuit is `nlayers` layers (layer 0 = output layer), its
 * gates concatenated in `gates` with per-layer counts.  The reference ties width to depth: layer i
 * reads a 2^(i+1)-entry layer (:166-178); other shapes hit its asserts (ZK_E_NVARS / ZK_E_NOT_POW2). */
typedef struct { uint64_t left, right, out, op; } zk_gate;
size_t zk_num_of_layer_variables(size_t layer_index);                               /* :166 */
size_t zk_wiring_index(size_t layer_index, size_t a, size_t b, size_t c);            /* convert_to_binary_and_to_decimal :180 */
size_t zk_circuit_eval_size(const zk_gate *gates, const size_t *gate_counts, size_t nlayers, size_t ninputs);
/* Circuit::evaluate :65-109 ; layer_sizes[nlayers+1], evals = layer 0 .. inputs concatenated */
int zk_circuit_evaluate(int field, const zk_gate *gates, const size_t *gate_counts, size_t nlayers,
                        const uint64_t *inputs, size_t ninputs, size_t *layer_sizes, uint64_t *evals);
/* add_i_and_mul_i_mle :126-163 : dense wiring predicates, built in HBM */
int zk_circuit_add_mul_mle(int field, const zk_gate *layer_gates, size_t ngates, size_t layer_index,
                           zk_table **add_i, zk_table **mul_i);
/* gkr_protocol::prove  gkr_protocol.rs:26-143.  Flattened Proof (:17-23):
 *   circuit_output[*output_len]; claimed_sum[1]; layer_claims[nlayers] (each layer's
 *   SumcheckProverProof.claimed_sum); coeffs: per layer rounds(L)*3 coefficients, rounds(L)=2(L+1);
 *   challenges: per layer rounds(L); wb_evals / wc_evals [nlayers-1].
 * A well-formed circuit (layer i writes wires 0 .. 2^i - 1, no gate twice) is proved from its gate lists (zk_gkr_sparse_*): the same transcript and
 * the same proof bytes without the 2^(3 i + 2)-entry dense predicates; the gate lists of the calling thread's last circuit stay compiled on the device.
 * ZK_GKR_DENSE_TABLES=1 (environment, read per call) keeps the reference's dense representation, which also serves every other shape. */
size_t zk_gkr_rounds(size_t layer_index);
int zk_gkr_prove(int field, const zk_gate *gates, const size_t *gate_counts, size_t nlayers,
                 const uint64_t *inputs, size_t ninputs, uint64_t *circuit_output, size_t *output_len,
                 uint64_t *claimed_sum, uint64_t *layer_claims, uint64_t *coeffs, uint64_t *challenges,
                 uint64_t *wb_evals, uint64_t *wc_evals);
/* gkr_protocol::verify :146-236 ; *ok = 1 / 0 */
int zk_gkr_verify(int field, const zk_gate *gates, const size_t *gate_counts, size_t nlayers,
                  const uint64_t *inputs, size_t ninputs, const uint64_t *circuit_output, size_t output_len,
                  const uint64_t *layer_claims, const uint64_t *coeffs, const uint64_t *wb_evals,
                  const uint64_t *wc_evals, int *ok);

/* ---- BLS12-381 G1 bases, MSM and multilinear KZG (multilinear_kzg/src) ---------------------------
 * Affine point = 12 u64: x[6] | y[6], Fq Montgomery limbs; x = y = 0 encodes infinity.  Group
 * results are affine-normalised: compare them as (x, y) (a projective triple is not unique). */
typedef struct zk_g1_bases zk_g1_bases;                 /* HBM-resident affine bases (TrustedSetup.g1_powers_of_tau, trusted_setup.rs:5-8) */
int zk_g1_bases_upload(const uint64_t *affine, size_t n, zk_g1_bases **out);
int zk_g1_bases_download(const zk_g1_bases *b, uint64_t *affine);
int zk_g1_bases_free(zk_g1_bases *b);
size_t zk_g1_bases_len(const zk_g1_bases *b);
/* synthetic benchmark bases P_i = [a + i*d] G, generated on the device */
int zk_g1_bases_synthetic(size_t n, const uint64_t *a_fr, const uint64_t *d_fr, zk_g1_bases **out);
int zk_g1_generator(uint64_t *out12);
int zk_g1_is_on_curve(const uint64_t *p12);
/* host-side group helpers (control path: combining per-GPU partial results, a handful of points) */
int zk_g1_add(const uint64_t *p12, const uint64_t *q12, uint64_t *out12);
int zk_g1_mul_fr(const uint64_t *p12, const uint64_t *scalar_fr, uint64_t *out12);

typedef struct {
    int window_bits, windows;
    uint64_t terms, entries, segments;
    float ms_digits, ms_sort, ms_buckets, ms_reduce, ms_total;    /* HIP-event times of the phases */
} zk_msm_stats;
/* sum_i [s_i] B_i by Pippenger (window_bits = 0: chosen from n; 2 .. 24).  scalars: Fr table of n terms
 * (n = bases length, any n >= 1).  This is the dot product of multilinear_kzg.rs:37-42 / :100-107.
 * Windows of more than 16 bits sort their entries most-significant-digit first (csrc/msm_sort_wide.cuh). */
int zk_msm_g1(const zk_table *scalars, const zk_g1_bases *bases, int window_bits, uint64_t *out12,
              zk_msm_stats *stats /* may be NULL */);

/* Optional, once per setup (TrustedSetup.g1_powers_of_tau is fixed across commits, trusted_setup.rs:5-8): keep one copy of the
 * points per window, 2^(c w) B_i, so that every window of a later zk_msm_g1 / zk_kzg_commit on these bases feeds ONE bucket set:
 * ceil(256 / c) bucket additions per term with c = 22 (window_bits = 0: chosen from n), one bucket reduction, no window
 * combination.  Costs ceil(256 / c) x 128 bytes per point of HBM and (W - 1) c doublings per point to build; the group element
 * returned by the MSM is the same.  A later call with another window size rebuilds the copies. */
int zk_g1_bases_precompute(zk_g1_bases *b, int window_bits);
int zk_g1_bases_precomputed_window(const zk_g1_bases *b);      /* 0: none */

/* compute_lagrange_basis  trusted_setup.rs:24-49 : eq table of tau, built in HBM (O(2^n)) */
int zk_kzg_lagrange_basis(const uint64_t *taus, size_t ntaus, zk_table **out);
/* compute_g1_powers_of_tau :51-60 : [L_i(tau)] G for all i (fixed-base windowed scalar mul) */
int zk_kzg_setup_g1(const uint64_t *taus, size_t ntaus, zk_g1_bases **out);
/* commit_to_polynomial  multilinear_kzg.rs:25-45 */
int zk_kzg_commit(const zk_table *poly, const zk_g1_bases *g1_powers, uint64_t *out12);
/* open_and_prove :50-126.  The reference runs a full-size naive dot product per round over the
 * blown-up quotient; the same group elements are obtained here as MSMs of sizes 2^(n-1) .. 1
 * against pre-summed bases (zk_kzg_opening_key, built once per setup).  n_g2 = the setup's
 * g2_powers_of_tau length (only compared, :60-64).  proofs: nopen affine points.
 * The independent level MSMs of one call run on up to 3 library-owned host threads, each with its
 * own stream (environment ZK_KZG_OPEN_THREADS = 1 .. 4; 1 keeps everything on the caller's thread
 * and stream); the call returns when all of them have finished. */
typedef struct zk_kzg_opening_key zk_kzg_opening_key;
int zk_kzg_opening_key_new(const zk_g1_bases *g1_powers, zk_kzg_opening_key **out);
int zk_kzg_opening_key_free(zk_kzg_opening_key *k);
/* Optional, once per key: zk_g1_bases_precompute on every pre-summed level of at least `min_points` points (0: 2^18), so that
 * the level MSMs of later openings (multilinear_kzg.rs:96-107) run on one bucket set each.  window_bits = 0: chosen per level.
 * Costs as zk_g1_bases_precompute, summed over the levels (about as much again as for the setup itself); the proofs are the
 * same group elements. */
int zk_kzg_opening_key_precompute(zk_kzg_opening_key *k, int window_bits, size_t min_points);
int zk_kzg_open(const zk_table *poly, const zk_g1_bases *g1_powers, const zk_kzg_opening_key *key,
                const uint64_t *opening, size_t nopen, size_t n_g2, uint64_t *evaluation, uint64_t *proofs);

/* prove_succinct  gkr/src/succinct_gkr_protocol.rs:35-169 (BLS12-381 Fr): the GKR proof of
 * zk_gkr_prove plus commit(inputs) (:42-44) and the two openings at the last layer's rb / rc
 * (:154-157).  rb_proofs / rc_proofs: nlayers affine points each. */
int zk_gkr_prove_succinct(const zk_gate *gates, const size_t *gate_counts, size_t nlayers,
                          const uint64_t *inputs, size_t ninputs, const zk_g1_bases *g1_powers, size_t n_g2,
                          uint64_t *circuit_output, size_t *output_len, uint64_t *claimed_sum,
                          uint64_t *layer_claims, uint64_t *coeffs, uint64_t *challenges,
                          uint64_t *wb_evals, uint64_t *wc_evals, uint64_t *commitment12,
                          uint64_t *rb_evaluation, uint64_t *rb_proofs, uint64_t *rc_evaluation,
                          uint64_t *rc_proofs);

/* ---- verifier side of multilinear KZG: G2, pairings (HOST code: O(n) work on n + 2 points, csrc/pairing.h) ------------
 * G2 points: affine over Fq2, 24 limbs = x.c0, x.c1, y.c0, y.c1 (6 x u64 Montgomery each), all zero = infinity.
 * GT elements: 72 limbs = the Fq2 coefficients of w^0 .. w^5 in Fq12 = Fq2[w]/(w^6 - (1 + u)), (c0, c1) each. */
int zk_g2_generator(uint64_t *out24);
int zk_g2_is_on_curve(const uint64_t *p24);                                 /* 1 / 0 */
int zk_g2_add(const uint64_t *p24, const uint64_t *q24, uint64_t *out24);
int zk_g2_mul_fr(const uint64_t *p24, const uint64_t *scalar_fr, uint64_t *out24);          /* mul_bigint(into_bigint) */
int zk_pairing(const uint64_t *g1_12, const uint64_t *g2_24, uint64_t *gt72);               /* P::pairing */
int zk_pairing_product_is_one(const uint64_t *g1s, const uint64_t *g2s, size_t n, int *ok);
/* compute_g2_powers_of_tau  trusted_setup.rs:62-72 : out[i] = [tau_i] G2 (ntaus x 24 limbs) */
int zk_kzg_setup_g2(const uint64_t *taus, size_t ntaus, uint64_t *out);
/* MultilinearKZG::verify  multilinear_kzg.rs:131-158 ; *ok = 1 / 0.  nopen != nproofs -> ZK_E_KZG_LEN (:137-141), as is
 * ng2 > nproofs (the reference indexes proofs[i] for every G2 power, :149-154) */
int zk_kzg_verify(const uint64_t *commitment12, const uint64_t *opening_values, size_t nopen, const uint64_t *evaluation,
                  const uint64_t *proofs, size_t nproofs, const uint64_t *g2_powers, size_t ng2, int *ok);

/* verify_succinct  gkr/src/succinct_gkr_protocol.rs:172-285 (BLS12-381 Fr): GKR verification without the inputs, then the
 * two KZG openings of the committed input polynomial at the last layer's challenges */
int zk_gkr_verify_succinct(const zk_gate *gates, const size_t *gate_counts, size_t nlayers, const uint64_t *circuit_output,
                           size_t output_len, const uint64_t *layer_claims, const uint64_t *coeffs, const uint64_t *wb_evals,
                           const uint64_t *wc_evals, const uint64_t *commitment12, const uint64_t *rb_evaluation,
                           const uint64_t *rb_proofs, size_t n_rb_proofs, const uint64_t *rc_evaluation, const uint64_t *rc_proofs,
                           size_t n_rc_proofs, const uint64_t *g2_powers, size_t ng2, int *ok);

/* ---- sparse (linear-time) GKR prover: the generalisation BASELINE config 4 needs -------------------------
 * The reference materialises dense wiring predicates add_i / mul_i of 2^(3i+2) entries and a dense f(b,c) of
 * 2^(2i+2) entries (arithmetic_circuit.rs:126-163, utils.rs:8-21) and ties a layer's width to its index
 * (:166-178).  Here a layer is a gate LIST (out, left, right, op) with its own widths: layer l has
 * 2^out_bits[l] outputs and reads the 2^out_bits[l+1] wires of the next layer (the inputs for the last one).
 * Each layer's 2k-round sumcheck is run in two phases (b then c) on tables built from the gate list in
 * O(#gates + 2^k):   phase 1  f = W(b) H1(b) + H0(b),   phase 2  f = A(c)(u + W(c)) + M(c) u W(c).
 * The round polynomials are the SAME polynomials as the dense definition's, so on circuits of the
 * reference's shape the proof is bit-identical to zk_gkr_prove / gkr_protocol::prove (tests/test_gpu_gkr_sparse.py).
 * With out_bits[0] = k0 > 1 the output claim uses k0 successive challenges (the reference has k0 = 1).
 * Proof layout as zk_gkr_prove with rounds(l) = 2 * in_bits(l). */
int zk_gkr_sparse_prove(int field, const zk_gate *gates, const size_t *gate_counts, size_t nlayers,
                        const uint32_t *out_bits, const uint64_t *inputs, size_t ninputs,
                        uint64_t *circuit_output, uint64_t *claimed_sum, uint64_t *layer_claims,
                        uint64_t *coeffs, uint64_t *challenges, uint64_t *wb_evals, uint64_t *wc_evals,
                        uint64_t *output_challenges /* out_bits[0] elements */, float *ms_layers /* nlayers, may be NULL */);
/* a circuit compiled once (gate lists uploaded and grouped by left / right / output index) and reused by
 * any number of proofs: the per-circuit preprocessing is O(#gates) host work that does not belong in a proof */
typedef struct zk_sparse_circuit zk_sparse_circuit;
int zk_sparse_circuit_new(const zk_gate *gates, const size_t *gate_counts, size_t nlayers, const uint32_t *out_bits,
                          size_t ninputs, zk_sparse_circuit **out);
int zk_sparse_circuit_free(zk_sparse_circuit *c);
int zk_gkr_sparse_prove_compiled(int field, const zk_sparse_circuit *c, const uint64_t *inputs, size_t ninputs,
                                 uint64_t *circuit_output, uint64_t *claimed_sum, uint64_t *layer_claims,
                                 uint64_t *coeffs, uint64_t *challenges, uint64_t *wb_evals, uint64_t *wc_evals,
                                 uint64_t *output_challenges, float *ms_layers);
/* independent evaluation of the wiring predicates at a point (the verifier's O(#gates) work):
 * add_r = sum_{add gates} w_g eq(rb, left_g) eq(rc, right_g), same for mul, with
 * w_g = alpha eq(pa, out_g) + beta eq(pb, out_g)  (layer 0: alpha = 1, beta = 0, pa = output challenges). */
int zk_gkr_sparse_wiring_eval(int field, const zk_gate *layer_gates, size_t ngates, uint32_t out_bits, uint32_t in_bits,
                              const uint64_t *alpha, const uint64_t *pa, const uint64_t *beta, const uint64_t *pb,
                              const uint64_t *rb, const uint64_t *rc, uint64_t *add_r, uint64_t *mul_r);
/* layer-by-layer evaluation of a sparse circuit on the GPU; evals = layer 0 .. inputs concatenated */
int zk_sparse_circuit_evaluate(int field, const zk_gate *gates, const size_t *gate_counts, size_t nlayers,
                               const uint32_t *out_bits, const uint64_t *inputs, size_t ninputs, uint64_t *evals);

/* ---- multi-GPU provers: one process per GPU, RCCL over xGMI -------------------------------------
 * A zk_comm is this rank's end of the node's communicator.  zk_comm_init_rccl creates an RCCL communicator on the calling
 * thread's device (ncclCommInitRank; librccl.so.1 is opened at first use): rank 0 calls zk_comm_unique_id and ships the 128
 * bytes to the other ranks out of band (file, MPI, torch.distributed store ...).  Every collective of the provers below is
 * enqueued on the calling thread's current stream between the kernels it separates -- the device never waits for the host
 * inside a sumcheck.  zk_comm_from_host_ops is the same interface over caller-supplied HOST-memory exchange callbacks (the
 * library stages device <-> pinned host and synchronises around each call): for hosts that own another transport, and for
 * the multi-rank tests on one GPU (RCCL refuses two ranks on one device).
 * The tables of the reference's provers shard by the LOW index bits: rank g of G = 2^k holds the entries i == g (mod G) as a
 * contiguous local table (local index i >> k), so that every round that folds variable 0 (prover.rs:62,
 * sumcheck_gkr_protocol.rs:57) is local; MSM terms shard by contiguous slices. */
typedef struct zk_comm zk_comm;
typedef struct {
    void *ctx;
    /* in-place element-wise sum over the ranks of `count` int64 words */
    int (*all_reduce_sum_i64)(void *ctx, int64_t *host_buf, size_t count);
    /* recv = send buffers of all ranks in rank order (bytes each) */
    int (*all_gather)(void *ctx, const void *host_send, void *host_recv, size_t bytes);
    /* root's recv = send buffers of all ranks in rank order; recv is NULL on the other ranks */
    int (*gather)(void *ctx, const void *host_send, void *host_recv, size_t bytes, int root);
    int (*broadcast)(void *ctx, void *host_buf, size_t bytes, int root);
} zk_comm_host_ops;                                               /* every callback returns 0 on success */
int zk_comm_unique_id(uint8_t out128[128]);                       /* ncclGetUniqueId */
int zk_comm_init_rccl(const uint8_t id128[128], int nranks, int rank, zk_comm **out);
int zk_comm_from_host_ops(const zk_comm_host_ops *ops, int nranks, int rank, zk_comm **out);
/* The ranks as THREADS of one process (one thread per GPU, or several ranks sharing one GPU in tests and rehearsals): a
 * group of `nranks` ends that exchange through the process's own host memory.  Every rank's thread creates its end with
 * zk_comm_from_local_group (backend "local-threads") and calls the provers concurrently, each on its own stream; the group
 * outlives its ends and is freed by the caller after them.  zk_comm_local_group_abort wakes every rank that waits in an
 * exchange (they return ZK_E_COMM): what a rank calls when its own proof failed, so that nobody hangs. */
typedef struct zk_comm_local_group zk_comm_local_group;
int zk_comm_local_group_new(int nranks, zk_comm_local_group **out);
int zk_comm_local_group_free(zk_comm_local_group *g);
int zk_comm_local_group_abort(zk_comm_local_group *g);
int zk_comm_from_local_group(zk_comm_local_group *g, int rank, zk_comm **out);
int zk_comm_free(zk_comm *c);
int zk_comm_rank(const zk_comm *c);
int zk_comm_size(const zk_comm *c);
const char *zk_comm_backend(const zk_comm *c);                    /* "rccl", "host-ops" or "local-threads" */
/* payload bytes this rank has received through the communicator so far (all-reduce: the buffer; all-gather / gather at the
 * root: the other ranks' parts; broadcast: the buffer on non-root ranks) and the number of collectives issued */
int zk_comm_stats(const zk_comm *c, uint64_t *bytes_received, uint64_t *collectives);
/* primitives, on device buffers, enqueued on the current stream (what the provers below are made of) */
int zk_comm_all_reduce_sum_i64(zk_comm *c, void *dev_buf, size_t count);
int zk_comm_all_gather(zk_comm *c, const void *dev_send, void *dev_recv, size_t bytes);
int zk_comm_broadcast(zk_comm *c, void *dev_buf, size_t bytes, int root);
/* the same exchanges on HOST buffers, for the callback kinds only ("host-ops", "local-threads"; ZK_E_ARG on an RCCL
 * communicator): op 0 = all-reduce(SUM) of n int64 words in place, 1 = all-gather of n bytes per rank into host_recv,
 * 2 = gather to `root` (host_recv on the root only), 3 = broadcast of n bytes from `root`.  Needs no device. */
int zk_comm_host_exchange(zk_comm *c, int op, void *host_buf, void *host_recv, size_t n, int root);

/* Prover::prove (prover.rs:35-71) of the global table whose low-bit shard is `shard` (local length 2^m, global 2^(m+k)).
 * Same proof bytes on every rank as zk_sumcheck_basic_prove on the interleaved table.  Per local round: one fused kernel,
 * ONE all-reduce of 18 int64 words, the transcript step on every rank's device.  absorb_table != 0 hashes the whole table
 * first (:38-39): the ranks stream their canonical bytes to rank 0 in chunks (gather), rank 0 absorbs them in global index
 * order and broadcasts the 208-byte sponge (25 lanes + fill) -- non-root ranks receive 208 bytes for the absorb. */
int zk_sharded_sumcheck_basic_prove(zk_comm *c, const zk_table *shard, int absorb_table, uint64_t *claimed_sum,
                                    uint64_t *round_polys /* (m + k) x 2 */, uint64_t *challenges /* m + k, may be NULL */);
/* sumcheck_gkr_protocol::prove (:24-67) on low-bit shards of the nprod x nfac tables; `t` must hold the same state on every
 * rank and is advanced identically.  final_values (nprod * nfac elements, may be NULL) = the fully folded tables. */
int zk_sharded_sumcheck_gkr_prove(zk_comm *c, const zk_table *const *shards, size_t nprod, size_t nfac,
                                  const uint64_t *claimed_sum, zk_transcript *t, uint64_t *round_coeffs, uint64_t *challenges,
                                  uint64_t *final_values);
/* MultilinearPolynomial::evaluate (evaluation_form.rs:21-33) of the sharded table at nvalues = m + k points: m local fold
 * rounds, one all-gather of G elements, k replicated rounds */
int zk_sharded_mle_evaluate(zk_comm *c, const zk_table *shard, const uint64_t *values, size_t nvalues, uint64_t *out);
/* commit_to_polynomial (multilinear_kzg.rs:37-42) with the terms sliced over the ranks: one Pippenger per rank, one
 * all-gather of G affine points (96 B each), G - 1 additions.  Same point on every rank. */
int zk_sharded_msm_g1(zk_comm *c, const zk_table *scalars_slice, const zk_g1_bases *bases_slice, int window_bits,
                      uint64_t *out12, zk_msm_stats *stats /* this rank's local MSM, may be NULL */);
/* open_and_prove (multilinear_kzg.rs:50-126) of the low-bit-sharded table (local length 2^m, G = 2^k ranks, nopen = m + k opening
 * values).  bases_local = the rank's own powers P_{j G + g} (the same low-bit shard of g1_powers_of_tau), key_local = the opening key
 * of bases_local (zk_kzg_opening_key_new; NULL: built for the call).  The first m proofs are sums over the ranks of local MSMs, the
 * last k are computed replicated from the G leftover entries and the G per-rank base totals: ONE all-gather of (m + 1) points + one
 * element per rank.  evaluation and the m + k proofs are the single-device ones, on every rank. */
int zk_sharded_kzg_open(zk_comm *c, const zk_table *shard, const zk_g1_bases *bases_local, const zk_kzg_opening_key *key_local,
                        const uint64_t *opening, size_t nopen, uint64_t *evaluation, uint64_t *proofs /* (m + k) x 12 */);

#ifdef __cplusplus
}
#endif
#endif
