// sumcheck_kernels.cuh -- HIP kernels for the degree-d sumcheck over a SumPolynomial
// (sum of NPROD products of NFAC multilinear tables), the GKR round polynomial.
//
// Reference (per round): generate_round_univariate sumcheck_gkr_protocol.rs:113-143 makes
// (NFAC+1) x NPROD x NFAC separate fold passes + element-wise product / sum passes, then
// SumPolynomial::partial_evaluate (sum_polynomial.rs:40-53) folds every table again.
// Here one kernel per round streams every table once:
//   round_evals_kernel        evals e(t) = sum_i sum_p prod_f X_t[p][f][i],  X_t = lo + t (hi - lo),
//                             t = 0..NFAC  (t-steps are additions only).  Products of TWO factors (every GKR round) use the
//                             nodes 0, 1 and infinity instead (r3): e(inf) = the X^2 coefficient = sum prod_f (hi - lo): one
//                             subtraction per factor where the node 2 took two operations, and nothing at all for a product
//                             whose second factor is constant (its X^2 coefficient is zero).  The reference interpolates the
//                             round polynomial from evaluations (sumcheck_gkr_protocol.rs:46-50) and sends COEFFICIENTS: the
//                             same polynomial from other nodes is the same message (kSecondNodeInfinity, zkmle_sumcheck.hip
//                             sumcheck_basis).
//   fold_round_evals_kernel   folds every table by the previous challenge AND produces the next
//                             round's e(t) from the folded values in the same pass.
// Algorithmic traffic of the fused kernel on NT = NPROD*NFAC tables of 2^m entries:
//   NT * 32 B * 2^m read + NT * 32 B * 2^(m-1) written; NT*2^(m-1) fold multiplications +
//   (NFAC+1)(NFAC-1) NPROD 2^(m-2) evaluation multiplications.
#pragma once
#include "mle_kernels.cuh"

namespace zk {

constexpr int kMaxProducts = 8;
constexpr int kMaxFactors = 3;

struct SumPolyTables {
    const void *in[kMaxProducts * kMaxFactors];   // [p * nfac + f]; nullptr = the CONSTANT factor cval[p] (second factor of a two-factor product)
    void *out[kMaxProducts * kMaxFactors];
    uint32_t cval[kMaxProducts][12];               // stored (Montgomery) form
    const void *cptr[kMaxProducts];                // non-null: the constant is read from device memory instead (one element; a value an
                                                   // earlier kernel of the same stream produced, e.g. u = W(rb*) of the sparse GKR prover)
};
// The round's exchange run by the LAST workgroup of the kernel that produces the round's evaluations (host-assisted transcript step only;
// dev_transcript.cuh round_finish_in_producer; null counter: the workgroups leave partials for a finish kernel instead).  Same scheme as the
// basic sumcheck's passes (basic_multi.cuh): every workgroup adds its sums to accumulator words on lines of their own with device-scope
// atomics and arrives at a two-level counter; the last one reads the npts sums, posts them and waits for the challenge.
struct HostMailbox;
struct RoundFin {
    unsigned *counter;       // the kernel's arrival counter, then one per group of `group` workgroups, 16 words apart; zero before the launch and left zero
    uint64_t *acc;           // npts x N words, kMultiAccStride words apart, zero before the launch and left zero
    int npts;                // evaluations per round
    int skip1;               // the point 1 is not evaluated: its words are neither written nor read (the host derives it)
    unsigned group;          // workgroups per arrival group
    uint64_t *limbs_out;     // sharded tables: the npts sums leave as (N + 1) 32-bit limbs in 64-bit words for the all-reduce; nothing is posted
    HostMailbox *mb;
    uint64_t seq;
    void *proof;
    size_t chal_slot;
    uint32_t *exp_out;       // non-null: the challenge also leaves as a uniform multiplier (ufield.cuh UniMul, L x L words) for the fused round that folds by it
    size_t per2;             // non-zero: a TWO-round exchange (split2_round_kernel): the npts = 9 sums go out together, two challenges come back, to chal_slot and chal_slot + per2
};
// wave 0 of every workgroup calls this; lane t < npts holds the workgroup's sum of evaluation t in `tot` (stored form, fully reduced)
template <class F> __device__ __forceinline__ void round_finish_in_producer(const RoundFin &f, const Fe<F> &tot);   // dev_transcript.cuh

// A product whose second factor is a constant c is a linear term: sum_i c X(i).  Its table is never materialised, loaded, folded or
// stored (a constant folds to itself); the evaluation products use c directly.  The sparse GKR prover's phases are
// W H1 + H0 * 1 and C W + A * u (zkmle_gkr_sparse.hip): three streamed tables instead of four.
template <class F> __device__ __forceinline__ Fe<F> const_factor(const SumPolyTables &t, int p) {
    if (t.cptr[p]) return fe_load<F>(t.cptr[p], 0);
    Fe<F> e;
#pragma unroll
    for (int i = 0; i < F::N; i++) e.l[i] = t.cval[p][i];
    return e;
}

// accumulate the NFAC+1 evaluation terms of one product at one pair index
template <class F, int NFAC>
__device__ __forceinline__ void accumulate_terms(const Fe<F> (&lo)[NFAC], const Fe<F> (&hi)[NFAC], Wide<F> (&acc)[NFAC + 1], int skip1 = 0) {
    Fe<F> v[NFAC], d[NFAC];
#pragma unroll
    for (int f = 0; f < NFAC; f++) {
        v[f] = lo[f];
        d[f] = fe_sub<F>(hi[f], lo[f]);
    }
    if constexpr (NFAC == 2) {                               // nodes 0, 1, infinity (header)
        wide_add_fe<F>(acc[0], fe_mul<F>(lo[0], lo[1]));
        if (!skip1) wide_add_fe<F>(acc[1], fe_mul<F>(hi[0], hi[1]));
        wide_add_fe<F>(acc[2], fe_mul<F>(d[0], d[1]));
        return;
    }
#pragma unroll
    for (int t = 0; t <= NFAC; t++) {
        if (!(skip1 && t == 1)) {                            // the point 1 is derived from the running claim (dev_transcript.cuh)
            Fe<F> term = v[0];
#pragma unroll
            for (int f = 1; f < NFAC; f++) term = fe_mul<F>(term, v[f]);
            wide_add_fe<F>(acc[t], term);
        }
        if (t < NFAC) {
#pragma unroll
            for (int f = 0; f < NFAC; f++) v[f] = fe_add<F>(v[f], d[f]);   // X_{t+1} = X_t + (hi - lo)
        }
    }
}

// ---- products summed BEFORE their Montgomery reduction (degree-2 rounds of the 4-limb fields) --------------------------------------
// A round evaluation is a sum of products; sum_i mont(a_i, b_i) = mont-reduce(sum_i a_i b_i), so the lane accumulates the raw
// double-width products (L^2 multiply-adds each instead of 2 L^2, no conversion back to 32-bit limbs) and the workgroup reduces the
// total once.  The accumulator keeps 2 L + 1 limbs of 29 bits in 32-bit words; every product adds less than 2^29 to a limb, so the
// carries are propagated every 6 products.  Operands may be any representative below 2^(29 L - 3): the nodes 0, 1, infinity use lo, hi
// and hi - lo + 4 p without a modular reduction.
template <class F> struct ProdConsts;            // c1 = 2^(29 L + SH) mod p, c2 = 2^(58 L + SH) mod p, 29-bit limbs
template <> struct ProdConsts<Fr381> {
    static ZK_HD uint32_t c1(int i) { constexpr uint32_t t[9] = {0x1ffff72bu, 0x000046a7u, 0x1f5f3540u, 0x0ce3021cu, 0x118f3661u, 0x008176cbu, 0x054e487cu, 0x102e8190u, 0x001e092eu}; return t[i]; }
    static ZK_HD uint32_t c2(int i) { constexpr uint32_t t[9] = {0x0e3677f5u, 0x06441022u, 0x10fe35fdu, 0x1f3f5076u, 0x122ddf09u, 0x18425209u, 0x1cd6203au, 0x19a93b3bu, 0x00047c05u}; return t[i]; }
};
template <> struct ProdConsts<Bn254Fq> {
    static ZK_HD uint32_t c1(int i) { constexpr uint32_t t[9] = {0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u}; return t[i]; }
    static ZK_HD uint32_t c2(int i) { constexpr uint32_t t[9] = {0x1e46cb83u, 0x072a411eu, 0x0feb9db7u, 0x08e6e8f9u, 0x0746d786u, 0x0ae2ff90u, 0x01e5e885u, 0x0d1ba21au, 0x0027a08bu}; return t[i]; }
};
template <> struct ProdConsts<Bn254Fr> {
    static ZK_HD uint32_t c1(int i) { constexpr uint32_t t[9] = {0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u}; return t[i]; }
    static ZK_HD uint32_t c2(int i) { constexpr uint32_t t[9] = {0x16d37a7au, 0x08833f88u, 0x0b72dfe0u, 0x07e2bbadu, 0x097730eau, 0x0de737eau, 0x1ed2d8f6u, 0x03e94703u, 0x001e4f71u}; return t[i]; }
};
template <class F> struct LazyProducts { static constexpr bool value = false; };
template <> struct LazyProducts<Fr381> { static constexpr bool value = true; };
template <> struct LazyProducts<Bn254Fq> { static constexpr bool value = true; };
template <> struct LazyProducts<Bn254Fr> { static constexpr bool value = true; };

template <class F> struct ProdAcc {
    uint32_t l[2 * UParams<F>::L + 1];
};
constexpr int kProdCarryEvery = 6;
template <class F> __device__ __forceinline__ ProdAcc<F> prod_zero() {
    ProdAcc<F> a;
#pragma unroll
    for (int j = 0; j <= 2 * UParams<F>::L; j++) a.l[j] = 0;
    return a;
}
// acc += a * b (integers; limbs of a and b below 2^29)
template <class F> __device__ __forceinline__ void prod_accumulate(ProdAcc<F> &acc, const Ufe<F> &a, const Ufe<F> &b) {
    constexpr int L = UParams<F>::L;
    uint64_t T[2 * L - 1];
#pragma unroll
    for (int j = 0; j < 2 * L - 1; j++) T[j] = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
#pragma unroll
        for (int j = 0; j < L; j++) T[i + j] += (uint64_t)a.l[j] * b.l[i];
    }
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 2 * L - 1; j++) {
        uint64_t v = T[j] + c;
        acc.l[j] += (uint32_t)v & UMASK;
        c = v >> UB;
    }
    acc.l[2 * L - 1] += (uint32_t)c & UMASK;
    acc.l[2 * L] += (uint32_t)(c >> UB);
}
template <class F> __device__ __forceinline__ void prod_carry(ProdAcc<F> &acc) {
#pragma unroll
    for (int j = 0; j < 2 * UParams<F>::L; j++) {
        acc.l[j + 1] += acc.l[j] >> UB;
        acc.l[j] &= UMASK;
    }
}
// the three (or two) evaluation terms of one product of two factors at one pair index
// SKIP1 (the point 1 is derived from the running claim, dev_transcript.cuh kDerive1): TWO accumulators instead of three -- 19 VGPRs less
// across the loop, which is what keeps the first-round kernel under 128 VGPRs (4 waves per SIMD instead of 3; r3).  acc[0] = e(0),
// acc[NACC - 1] = e(infinity), acc[1] = e(1) when it is evaluated.  `const2`: the second factor is the constant (lo[1] = hi[1]): the product is
// linear in X, nothing to add to e(infinity).
template <class F, bool SKIP1>
__device__ __forceinline__ void accumulate_terms_lazy(const Fe<F> (&lo)[2], const Fe<F> (&hi)[2], ProdAcc<F> (&acc)[SKIP1 ? 2 : 3], bool const2) {
    const Ufe<F> l0 = u_from_limbs32<F>(lo[0]), l1 = u_from_limbs32<F>(lo[1]);
    const Ufe<F> h0 = u_from_limbs32<F>(hi[0]), h1 = u_from_limbs32<F>(hi[1]);
    prod_accumulate<F>(acc[0], l0, l1);
    if constexpr (!SKIP1) prod_accumulate<F>(acc[1], h0, h1);
    // X^2 coefficient: (hi - lo + 4 p)(hi' - lo' + 4 p).  With a constant second factor the second difference is 4 p and the product a multiple of p:
    // skipped where the skip is free (SKIP1: the variant every large round runs); the three-accumulator variant keeps the loop branch-free
    // (the branch cost it 13 VGPRs and its fourth wave per SIMD).
    if (!SKIP1 || !const2) prod_accumulate<F>(acc[SKIP1 ? 1 : 2], usub<F>(h0, l0), usub<F>(h1, l1));
}
// the same on operands that already are normalized 29-bit limbs (the folded pair as ufold leaves it: any representative below 4 p)
template <class F, bool SKIP1>
__device__ __forceinline__ void accumulate_terms_lazy_u(const Ufe<F> (&lo)[2], const Ufe<F> (&hi)[2], ProdAcc<F> (&acc)[SKIP1 ? 2 : 3], bool const2) {
    prod_accumulate<F>(acc[0], lo[0], lo[1]);
    if constexpr (!SKIP1) prod_accumulate<F>(acc[1], hi[0], hi[1]);
    if (!SKIP1 || !const2) prod_accumulate<F>(acc[SKIP1 ? 1 : 2], usub<F>(hi[0], lo[0]), usub<F>(hi[1], lo[1]));
}
// The pass's challenge as a uniform multiplier (ufield.cuh UniMul) in scalar registers.  `rexp` (81 words, written by whoever produced
// the challenge: dev_transcript.cuh challenge_expand) is read with uniform loads; without it (and without a kernel argument) lane i < L of every wave works out row i
// from the challenge itself (one product per wave) and the rows are read back lane by lane.
// (a launch whose challenge the HOST knows -- zk_sumpoly_fold_round_evals -- carries the rows as a kernel argument)
struct UniArg {
    uint32_t valid;
    uint32_t t[81];
};
template <class F> __device__ __forceinline__ void unimul_load(UniMul<F> &m, const uint32_t *__restrict__ rexp, const UniArg &ua, const Fe<F> &r) {
    constexpr int L = UParams<F>::L;
    if (rexp) {
#pragma unroll
        for (int i = 0; i < L; i++) {
#pragma unroll
            for (int j = 0; j < L; j++) m.t[i][j] = __builtin_amdgcn_readfirstlane(rexp[i * L + j]);
        }
        return;
    }
    if (ua.valid) {
#pragma unroll
        for (int i = 0; i < L; i++) {
#pragma unroll
            for (int j = 0; j < L; j++) m.t[i][j] = ua.t[(i * L + j) % 81];
        }
        return;
    }
    const unsigned lane = threadIdx.x & 63u;
    const Ufe<F> row = unimul_row<F>(r, lane < (unsigned)L ? (int)lane : 0);
#pragma unroll
    for (int i = 0; i < L; i++) {
#pragma unroll
        for (int j = 0; j < L; j++) m.t[i][j] = __builtin_amdgcn_readlane(row.l[j], i);
    }
}
// 2 L + 1 normalized 29-bit limbs -> 2 N + 2 saturated 32-bit words
template <class F> struct ProdWide {
    uint32_t l[2 * F::N + 2];
};
template <class F> __device__ __forceinline__ ProdWide<F> prod_to_wide(const ProdAcc<F> &a) {
    constexpr int NL = 2 * UParams<F>::L + 1, NW = 2 * F::N + 2;
    ProdWide<F> r;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const int q = 32 * w, i = q / UB, off = q - UB * i;
        uint64_t v = i < NL ? (uint64_t)a.l[i] >> off : 0;
        const int have = UB - off;
        if (i + 1 < NL) v |= (uint64_t)a.l[i + 1] << have;
        if (have + UB < 32 && i + 2 < NL) v |= (uint64_t)a.l[i + 2] << (have + UB);
        r.l[w] = (uint32_t)v;
    }
    return r;
}
template <class F> __device__ __forceinline__ void prodwide_add(ProdWide<F> &w, const ProdWide<F> &o) {
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 2 * F::N + 2; i++) w.l[i] = __builtin_addc(w.l[i], o.l[i], c, &c);
}
template <class F, int CTRL, int ROW_MASK> __device__ __forceinline__ void prodwide_dpp_step(ProdWide<F> &v) {
    ProdWide<F> o;
#pragma unroll
    for (int k = 0; k < 2 * F::N + 2; k++) o.l[k] = dpp_or_zero<CTRL, ROW_MASK>(v.l[k]);
    prodwide_add<F>(v, o);
}
// S 2^(-32 N) mod p for a sum S of raw products (2 N + 2 words): S = A + B 2^(29 L) + C 2^(58 L), each part times its constant
template <class F> __device__ __forceinline__ Fe<F> prodwide_reduce(const ProdWide<F> &w) {
    constexpr int L = UParams<F>::L, NW = 2 * F::N + 2;
    Ufe<F> part[3], cst[3];
#pragma unroll
    for (int t = 0; t < 3; t++) {
#pragma unroll
        for (int j = 0; j < L; j++) {
            const int q = UB * (t * L + j), wi = q >> 5, off = q & 31;
            uint32_t lo_w = wi < NW ? w.l[wi] : 0u, hi_w = (wi + 1) < NW ? w.l[wi + 1] : 0u;
            uint32_t v = off ? ((lo_w >> off) | (hi_w << (32 - off))) : lo_w;
            part[t].l[j] = v & UMASK;
        }
    }
#pragma unroll
    for (int j = 0; j < L; j++) {
        cst[0].l[j] = j == 0 ? (1u << UParams<F>::SH) : 0u;
        cst[1].l[j] = ProdConsts<F>::c1(j);
        cst[2].l[j] = ProdConsts<F>::c2(j);
    }
    Ufe<F> sum = uadd<F>(uadd<F>(umul<F>(part[0], cst[0]), umul<F>(part[1], cst[1])), umul<F>(part[2], cst[2]));   // < 6 p
    Ufe<F> ru;
#pragma unroll
    for (int j = 0; j < L; j++) ru.l[j] = UParams<F>::r_u(j);
    return u_to_limbs32<F>(u_reduce_once<F>(umul<F>(sum, ru)));           // x 2^(29 L) / 2^(29 L): the same value, below 2 p
}
// workgroup totals of the lazily accumulated products -> partials, same layout as write_partials
template <class F, bool SKIP1>
__device__ __forceinline__ void write_partials_lazy(ProdAcc<F> (&acc)[SKIP1 ? 2 : 3], ProdWide<F> *sh, void *partials, const RoundFin &fin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    constexpr int skip1 = SKIP1 ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (k == 1 && SKIP1) continue;
        ProdAcc<F> &a = acc[(SKIP1 && k == 2) ? 1 : k];
        prod_carry<F>(a);
        ProdWide<F> v = prod_to_wide<F>(a);
        prodwide_dpp_step<F, 0x111, 0xf>(v);
        prodwide_dpp_step<F, 0x112, 0xf>(v);
        prodwide_dpp_step<F, 0x114, 0xf>(v);
        prodwide_dpp_step<F, 0x118, 0xf>(v);
        prodwide_dpp_step<F, 0x142, 0xa>(v);
        prodwide_dpp_step<F, 0x143, 0xc>(v);
        if (lane == 63) sh[k * nw + wave] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x >= (fin.counter ? 64 : 3)) return;
    Fe<F> tot = fe_zero<F>();
    if (threadIdx.x < 3 && !(threadIdx.x == 1 && skip1)) {
        ProdWide<F> t = sh[threadIdx.x * nw];
        for (int w = 1; w < nw; w++) prodwide_add<F>(t, sh[threadIdx.x * nw + w]);
        tot = prodwide_reduce<F>(t);
    }
    if (fin.counter) round_finish_in_producer<F>(fin, tot);
    else fe_store<F>(partials, (size_t)threadIdx.x * gridDim.x + blockIdx.x, tot);
}

template <class F, int NFAC>
__device__ __forceinline__ void write_partials(Wide<F> (&acc)[NFAC + 1], Wide<F> *sh, void *partials, const RoundFin &fin) {
    Fe<F> tot = fe_zero<F>();
    const bool have = block_reduce_wide<F, NFAC + 1>(acc, sh, tot);
    if (fin.counter) {
        if (threadIdx.x < 64) round_finish_in_producer<F>(fin, tot);
    } else if (have) fe_store<F>(partials, (size_t)threadIdx.x * gridDim.x + blockIdx.x, tot);
}

// tables of `2 * half` entries; partials[t * gridDim.x + block]
template <class F, int NFAC, bool SKIP1 = false>
__global__ void __launch_bounds__(kBlock) round_evals_kernel(SumPolyTables tabs, int nprod, size_t half, void *__restrict__ partials, RoundFin fin) {
    if constexpr (NFAC == 2 && LazyProducts<F>::value) {
        __shared__ ProdWide<F> shp[3 * kBlock / 64];
        constexpr int NACC = SKIP1 ? 2 : 3;
        ProdAcc<F> pacc[NACC];
#pragma unroll
        for (int t = 0; t < NACC; t++) pacc[t] = prod_zero<F>();
        int pending = 0;
        const size_t pstride = (size_t)gridDim.x * blockDim.x;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += pstride) {
            for (int p = 0; p < nprod; p++) {
                Fe<F> lo[2], hi[2];
#pragma unroll
                for (int f = 0; f < 2; f++) {
                    if (f == 1 && tabs.in[p * 2 + 1] == nullptr) { lo[1] = hi[1] = const_factor<F>(tabs, p); continue; }
                    lo[f] = fe_load<F>(tabs.in[p * 2 + f], i);
                    hi[f] = fe_load<F>(tabs.in[p * 2 + f], i + half);
                }
                accumulate_terms_lazy<F, SKIP1>(lo, hi, pacc, tabs.in[p * 2 + 1] == nullptr);
                if (++pending == kProdCarryEvery) {
                    pending = 0;
#pragma unroll
                    for (int t = 0; t < NACC; t++) prod_carry<F>(pacc[t]);
                }
            }
        }
        write_partials_lazy<F, SKIP1>(pacc, shp, partials, fin);
        return;
    }
    __shared__ Wide<F> sh[(NFAC + 1) * kBlock / 64];
    Wide<F> acc[NFAC + 1];
#pragma unroll
    for (int t = 0; t <= NFAC; t++) acc[t] = wide_zero<F>();
    size_t stride = (size_t)gridDim.x * blockDim.x;
    const int skip1 = SKIP1 ? 1 : 0;
    (void)skip1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += stride) {
        for (int p = 0; p < nprod; p++) {
            Fe<F> lo[NFAC], hi[NFAC];
#pragma unroll
            for (int f = 0; f < NFAC; f++) {
                if (NFAC == 2 && f == 1 && tabs.in[p * NFAC + 1] == nullptr) { lo[1] = hi[1] = const_factor<F>(tabs, p); continue; }
                lo[f] = fe_load<F>(tabs.in[p * NFAC + f], i);
                hi[f] = fe_load<F>(tabs.in[p * NFAC + f], i + half);
            }
            accumulate_terms<F, NFAC>(lo, hi, acc);
        }
    }
    write_partials<F, NFAC>(acc, sh, partials, fin);
}

// tables of 4q entries in, 2q out; lane i folds outputs i and i+q of every table, then uses them
// as the (lo, hi) pair of the NEXT round.
template <class F, int NFAC, bool SKIP1>
__device__ __forceinline__ void fold_round_evals_body(const SumPolyTables &tabs, int nprod, size_t q, const Fe<F> &r, void *__restrict__ partials,
                                                      const void *__restrict__ rp, const RoundFin &fin, const uint32_t *__restrict__ rexp, const UniArg &ua) {
    constexpr int skip1 = SKIP1 ? 1 : 0;
    if constexpr (NFAC == 2 && LazyProducts<F>::value) {
        __shared__ ProdWide<F> shp[3 * kBlock / 64];
        constexpr int NACC = SKIP1 ? 2 : 3;
        ProdAcc<F> pacc[NACC];
#pragma unroll
        for (int t = 0; t < NACC; t++) pacc[t] = prod_zero<F>();
        int pending = 0;
        const size_t pstride = (size_t)gridDim.x * blockDim.x;
        UniMul<F> um;                                            // the challenge as a uniform multiplier: 99 multiply-adds per fold instead of 162 (ufield.cuh)
        unimul_load<F>(um, rexp, ua, challenge_arg<F>(r, rp));
        // The four elements of a table are requested together: left alone the scheduler requests two, folds, requests the other two -- four
        // loads in flight per wave at three waves per SIMD (r4: 170 -> 166 us on 4 x 2^22).  Requesting the NEXT table ahead as well costs a wave
        // (196 VGPRs) and loses: 174 us (tools/microbench_round.hip).
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += pstride) {
            for (int p = 0; p < nprod; p++) {
                Ufe<F> lo[2], hi[2];
#pragma unroll
                for (int f = 0; f < 2; f++) {
                    const void *src = tabs.in[p * 2 + f];
                    void *dst = tabs.out[p * 2 + f];
                    if (f == 1 && src == nullptr) { lo[1] = hi[1] = u_from_limbs32<F>(const_factor<F>(tabs, p)); continue; }
                    Fe<F> a0 = fe_load<F>(src, i), a1 = fe_load<F>(src, i + q);
                    Fe<F> b0 = fe_load<F>(src, i + 2 * q), b1 = fe_load<F>(src, i + 3 * q);
                    __builtin_amdgcn_sched_barrier(0);
                    lo[f] = ufold<F>(um, u_from_limbs32<F>(a0), u_from_limbs32<F>(b0));
                    hi[f] = ufold<F>(um, u_from_limbs32<F>(a1), u_from_limbs32<F>(b1));
                    fe_store<F>(dst, i, fe_from_u_below_2p<F>(lo[f]));
                    fe_store<F>(dst, i + q, fe_from_u_below_2p<F>(hi[f]));
                }
                accumulate_terms_lazy_u<F, SKIP1>(lo, hi, pacc, tabs.in[p * 2 + 1] == nullptr);
                if (++pending == kProdCarryEvery) {
                    pending = 0;
#pragma unroll
                    for (int t = 0; t < NACC; t++) prod_carry<F>(pacc[t]);
                }
            }
        }
        write_partials_lazy<F, SKIP1>(pacc, shp, partials, fin);
        return;
    }
    __shared__ Wide<F> sh[(NFAC + 1) * kBlock / 64];
    Wide<F> acc[NFAC + 1];
#pragma unroll
    for (int t = 0; t <= NFAC; t++) acc[t] = wide_zero<F>();
    size_t stride = (size_t)gridDim.x * blockDim.x;
    const Multiplier<F> mr(challenge_arg<F>(r, rp));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += stride) {
        for (int p = 0; p < nprod; p++) {
            Fe<F> lo[NFAC], hi[NFAC];
#pragma unroll
            for (int f = 0; f < NFAC; f++) {
                const void *src = tabs.in[p * NFAC + f];
                void *dst = tabs.out[p * NFAC + f];
                if (NFAC == 2 && f == 1 && src == nullptr) { lo[1] = hi[1] = const_factor<F>(tabs, p); continue; }
                Fe<F> a0 = fe_load<F>(src, i), a1 = fe_load<F>(src, i + q);
                Fe<F> b0 = fe_load<F>(src, i + 2 * q), b1 = fe_load<F>(src, i + 3 * q);
                lo[f] = fe_add<F>(a0, mr.times(fe_sub<F>(b0, a0)));
                hi[f] = fe_add<F>(a1, mr.times(fe_sub<F>(b1, a1)));
                fe_store<F>(dst, i, lo[f]);
                fe_store<F>(dst, i + q, hi[f]);
            }
            accumulate_terms<F, NFAC>(lo, hi, acc, skip1);
        }
    }
    write_partials<F, NFAC>(acc, sh, partials, fin);
}
template <class F, int NFAC, bool SKIP1 = false>
__global__ void __launch_bounds__(kBlock) fold_round_evals_kernel(SumPolyTables tabs, int nprod, size_t q, Fe<F> r, void *__restrict__ partials,
                                                                  const void *__restrict__ rp, RoundFin fin, const uint32_t *__restrict__ rexp, UniArg ua) {
    fold_round_evals_body<F, NFAC, SKIP1>(tabs, nprod, q, r, partials, rp, fin, rexp, ua);
}
// The same round for SHORT tables of products of two factors.  Below ~2^15 pair indices the launch above does not fill the
// device and lasts as long as one lane's chain of products (14 for two products); here a wave takes ONE table for 64 consecutive
// pair indices (2 fold products per lane), the factors of a product meet in LDS, and the even wave of a product evaluates the
// points 0 and infinity, the odd one the point 1: a chain of 4.  blockDim = 64 * ntab (ntab <= 8), grid = q / 64 = the number of partials.
constexpr size_t kSplitRoundMaxQ = (size_t)1 << 15;
template <class F>
__global__ void __launch_bounds__(512) fold_round_evals_split_kernel(SumPolyTables tabs, size_t q, Fe<F> r, void *__restrict__ partials,
                                                                     const void *__restrict__ rp, int skip1, RoundFin fin) {
    __shared__ Fe<F> exch[2 * 64 * 8];
    __shared__ Wide<F> sh[3 * 8];
    const unsigned k = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const size_t i = (size_t)blockIdx.x * 64 + lane;           // q is a multiple of 64
    Wide<F> acc[3] = {wide_zero<F>(), wide_zero<F>(), wide_zero<F>()};
    const Multiplier<F> mr(challenge_arg<F>(r, rp));
    const void *src = tabs.in[k];
    void *dst = tabs.out[k];
    Fe<F> lo, hi;
    if (src == nullptr) {                                       // constant factor (wave-uniform): nothing to load, fold or store
        lo = hi = const_factor<F>(tabs, (int)(k >> 1));
    } else {
        Fe<F> a0 = fe_load<F>(src, i), a1 = fe_load<F>(src, i + q);
        Fe<F> b0 = fe_load<F>(src, i + 2 * q), b1 = fe_load<F>(src, i + 3 * q);
        lo = fe_add<F>(a0, mr.times(fe_sub<F>(b0, a0)));
        hi = fe_add<F>(a1, mr.times(fe_sub<F>(b1, a1)));
        fe_store<F>(dst, i, lo);
        fe_store<F>(dst, i + q, hi);
    }
    exch[2 * (k * 64 + lane)] = lo;
    exch[2 * (k * 64 + lane) + 1] = hi;
    __syncthreads();
    const Fe<F> lo2 = exch[2 * ((k ^ 1u) * 64 + lane)], hi2 = exch[2 * ((k ^ 1u) * 64 + lane) + 1];
    if ((k & 1u) == 0) {
        wide_add_fe<F>(acc[0], fe_mul<F>(lo, lo2));
        wide_add_fe<F>(acc[2], fe_mul<F>(fe_sub<F>(hi, lo), fe_sub<F>(hi2, lo2)));      // node infinity: the X^2 coefficient (header)
    } else if (!skip1) {
        wide_add_fe<F>(acc[1], fe_mul<F>(hi, hi2));
    }
    Fe<F> tot = fe_zero<F>();
    const bool have = block_reduce_wide<F, 3>(acc, sh, tot);
    if (fin.counter) {
        if (threadIdx.x < 64) round_finish_in_producer<F>(fin, tot);
    } else if (have) fe_store<F>(partials, (size_t)threadIdx.x * gridDim.x + blockIdx.x, tot);
}

// TWO rounds per launch and exchange on short tables of two products of two factors (the tail's scheme, dev_transcript.cuh sumcheck_tail_kernel, grid-wide):
// the tables are folded by the ONE challenge (rp1 == null: 8 qq entries in) or the TWO challenges (16 qq entries in) that are pending, to T of 4 qq entries,
// and the nine sums over the quads (T[i], T[i + qq] | T[i + 2 qq], T[i + 3 qq]) that carry the next two rounds leave in one exchange.
// Workgroup = 1024 lanes = 64 quads: wave (k, e) folds entry e of table k's quad (three folds at most), the entries meet in LDS, then wave w takes the
// (sum, product) pairs w and w + 16 of the 9 x 2.  Seven single split rounds and the tail's two longest rounds (131 us) become four of these.
template <class F>
__global__ void __launch_bounds__(1024) split2_round_kernel(SumPolyTables tabs, size_t qq, const void *__restrict__ rp0, const void *__restrict__ rp1, RoundFin fin) {
    __shared__ Fe<F> exch[16 * 64];                            // [(k * 4 + e) * 64 + lane]
    __shared__ Wide<F> part[18];
    const unsigned w = threadIdx.x >> 6, lane = threadIdx.x & 63u, k = w >> 2, e = w & 3u;
    const size_t i = (size_t)blockIdx.x * 64 + lane;           // qq is a multiple of 64
    const void *src = tabs.in[k];
    Fe<F> v;
    if (src == nullptr) {                                      // constant factor (wave-uniform)
        v = const_factor<F>(tabs, (int)(k >> 1));
    } else {
        const Multiplier<F> m0(fe_load<F>(rp0, 0));
        const size_t j = (size_t)e * qq + i;
        if (rp1 == nullptr) {
            const Fe<F> x = fe_load<F>(src, j), y = fe_load<F>(src, j + 4 * qq);
            v = fe_add<F>(x, m0.times(fe_sub<F>(y, x)));
        } else {
            const Fe<F> x0 = fe_load<F>(src, j), y0 = fe_load<F>(src, j + 8 * qq), x1 = fe_load<F>(src, j + 4 * qq), y1 = fe_load<F>(src, j + 12 * qq);
            const Fe<F> l0 = fe_add<F>(x0, m0.times(fe_sub<F>(y0, x0))), l1 = fe_add<F>(x1, m0.times(fe_sub<F>(y1, x1)));
            const Multiplier<F> m1(fe_load<F>(rp1, 0));
            v = fe_add<F>(l0, m1.times(fe_sub<F>(l1, l0)));
        }
        fe_store<F>(tabs.out[k], j, v);
    }
    exch[w * 64 + lane] = v;
    __syncthreads();
#pragma unroll 1
    for (unsigned t = w; t < 18; t += 16) {                    // (sum, product) pair t: sum t % 9 of product t / 9
        const unsigned kind = t % 9u, p = t / 9u;
        Fe<F> o[2];
#pragma unroll
        for (int f = 0; f < 2; f++) {
            const Fe<F> *q4 = exch + (size_t)((p * 2 + f) * 4) * 64 + lane;           // a, b, c, d at q4[0], q4[64], q4[128], q4[192]
            if (kind < 4) o[f] = q4[kind * 64];
            else if (kind < 8) {
                const unsigned hi = kind == 4 ? 2 : kind == 5 ? 3 : kind == 6 ? 1 : 3, lo = kind == 4 ? 0 : kind == 5 ? 1 : kind == 6 ? 0 : 2;
                o[f] = fe_sub<F>(q4[hi * 64], q4[lo * 64]);
            } else o[f] = fe_sub<F>(fe_sub<F>(q4[192], q4[128]), fe_sub<F>(q4[64], q4[0]));
        }
        Wide<F> acc[1] = {wide_zero<F>()};
        wide_add_fe<F>(acc[0], fe_mul<F>(o[0], o[1]));
        wave_reduce_wide<F, 1>(acc);
        if (lane == 63) part[t] = acc[0];
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    Fe<F> tot = fe_zero<F>();
    if (lane < 9) {
        Wide<F> s2 = part[lane];
        wide_add<F>(s2, part[9 + lane]);
        tot = wide_reduce<F>(s2);
    }
    round_finish_in_producer<F>(fin, tot);
}

// element-wise reduce of a SumPolynomial to one table: out[i] = sum_p prod_f X[p][f][i]
// (add_polynomials_element_wise sum_polynomial.rs:57-76 over multiply_polynomials_element_wise
//  product_polynomial.rs:58-73)
template <class F>
__global__ void sumpoly_reduce_kernel(SumPolyTables tabs, int nprod, int nfac, size_t len, void *__restrict__ out) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) {
        Fe<F> acc = fe_zero<F>();
        for (int p = 0; p < nprod; p++) {
            Fe<F> term = fe_load<F>(tabs.in[p * nfac], i);
            for (int f = 1; f < nfac; f++) term = fe_mul<F>(term, fe_load<F>(tabs.in[p * nfac + f], i));
            acc = fe_add<F>(acc, term);
        }
        fe_store<F>(out, i, acc);
    }
}

// dst[pos[k]] = one  (wiring predicates add_i / mul_i, arithmetic_circuit.rs:136-156)
template <class F>
__global__ void scatter_one_kernel(void *__restrict__ dst, const uint64_t *__restrict__ pos, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fe_store<F>(dst, pos[i], fe_one<F>());
}

}  // namespace zk
