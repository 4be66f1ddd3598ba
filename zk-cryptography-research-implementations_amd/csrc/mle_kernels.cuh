// mle_kernels.cuh -- HIP kernels for the multilinear-extension table operations (gfx950).
//
// All kernels stream AoS tables of Montgomery elements (32 B for 4x64-bit fields) with 16-byte
// vector loads, one element per lane per access, grid-stride; they are HBM-bound by design:
//   fold:  2 x 32 B read + 32 B write per field multiplication  (96 B / mul)
// Reference loops restated: polynomials/src/multilinear/evaluation_form.rs:61-106 (fold),
// :49-57 / :108-163 (element-wise and tensor ops), sumcheck_protocol/src/basic_sumcheck/prover.rs:74-89
// (half sums).
#pragma once
#include <stdlib.h>

#include "ufield.cuh"

namespace zk {

constexpr int kBlock = 256;          // 4 waves of 64
// Streaming kernels launch one element per lane (measured on MI355X, profiles/microbench_r1.jsonl:
// 2^23-output fold 5.28 TB/s at 32768 blocks vs 4.67 TB/s capped at 2048); reductions keep a
// bounded grid because they emit one partial per block.
constexpr int kMaxBlocks = 1 << 16;
constexpr int kMaxReduceBlocks = 16384;  // capacity of the partial buffers; the launch cap is reduce_block_cap()

// ---- synthetic data: SplitMix64 keyed by (seed, element index, word) -----------------
ZK_HD uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// element `idx` of the stream `seed`: N/2 pseudo-random u64 words, top limb masked to
// (bitlen(p) - 1) bits so the value is < p; the words ARE the stored (Montgomery) limbs.
template <class F> ZK_HD Fe<F> random_element(uint64_t seed, uint64_t idx) {
    Fe<F> e;
#pragma unroll
    for (int k = 0; k < F::N / 2; k++) {
        uint64_t w = splitmix64(seed ^ splitmix64(idx * (F::N / 2) + k));
        e.l[2 * k] = (uint32_t)w;
        e.l[2 * k + 1] = (uint32_t)(w >> 32);
    }
    uint32_t top = F::p(F::N - 1);
    int bits = 32 - __builtin_clz(top);          // bit length of the top limb of p
    e.l[F::N - 1] &= (bits >= 2) ? ((1u << (bits - 1)) - 1u) : 0u;
    return e;
}
template <class F> __global__ void fill_random_kernel(void *out, size_t len, uint64_t seed, size_t first, size_t step = 1) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride)
        fe_store<F>(out, i, random_element<F>(seed, first + i * step));
}

// a loop-invariant multiplier (the round challenge r): converted once per lane to the 29-bit form
template <class F> struct Multiplier {
    Ufe<F> u;
    __device__ __forceinline__ explicit Multiplier(const Fe<F> &r) : u(u_from_limbs32<F>(r)) {}
    __device__ __forceinline__ Fe<F> times(const Fe<F> &x) const { return fe_mul_u_pre<F>(u, x); }
};

// A challenge is passed by value (host-driven callers) or read from device memory (`rp`, written by the previous
// round's finish kernel: dev_transcript.cuh) -- a uniform 32-byte load.
template <class F> __device__ __forceinline__ Fe<F> challenge_arg(const Fe<F> &r, const void *rp) {
    return rp ? fe_load<F>(rp, 0) : r;
}

// ---- fold: out[i] = y1 + r * (y2 - y1)   evaluation_form.rs:88-89 -----------------------------------
// `power` = n - 1 - var (:80).  Output index i maps to y1 index j = i with a zero bit inserted
// at position `power` (the reference's j-walk :98-102), y2 = j | 1<<power (:82).
template <class F> __global__ void fold_kernel(const void *__restrict__ in, void *__restrict__ out,
                                              size_t half, unsigned power, Fe<F> r, const void *__restrict__ rp = nullptr) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t lowmask = ((size_t)1 << power) - 1;
    const Multiplier<F> mr(challenge_arg<F>(r, rp));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += stride) {
        size_t j = ((i & ~lowmask) << 1) | (i & lowmask);
        Fe<F> y1 = fe_load<F>(in, j);
        Fe<F> y2 = fe_load<F>(in, j | ((size_t)1 << power));
        fe_store<F>(out, i, fe_add<F>(y1, mr.times(fe_sub<F>(y2, y1))));
    }
}

// variable 0 (the only variable production callers fold): two contiguous input streams,
// one output element per lane, no index arithmetic
template <class F> __global__ void fold0_kernel(const void *__restrict__ in, void *__restrict__ out, size_t half, Fe<F> r) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const Multiplier<F> mr(r);
    Fe<F> y1 = fe_load<F>(in, i), y2 = fe_load<F>(in, i + half);
    fe_store<F>(out, i, fe_add<F>(y1, mr.times(fe_sub<F>(y2, y1))));
}

// evaluate()'s last folds in ONE launch: a table of <= 2 * kEvalTailBlock entries is folded by up to kEvalTailVars successive
// values by one workgroup (level 0 reads `in`, writes `tmp`; later levels fold `tmp` in place: thread i only overwrites the entry
// it alone reads).  A chain of 11 tiny launches is ~60 us of pure launch latency otherwise.
constexpr int kEvalTailBlock = 1024;
constexpr int kEvalTailVars = 11;
template <class F> struct EvalTailValues {
    Fe<F> r[kEvalTailVars];
};
template <class F> __global__ void __launch_bounds__(kEvalTailBlock) evaluate_tail_kernel(const void *__restrict__ in, void *__restrict__ tmp, size_t len,
                                                                                            EvalTailValues<F> vals, int nvals) {
    const void *src = in;
    for (int k = 0; k < nvals; k++) {
        const size_t half = len >> 1;
        if (threadIdx.x < half) {
            const Multiplier<F> mr(vals.r[k]);
            Fe<F> y1 = fe_load<F>(src, threadIdx.x), y2 = fe_load<F>(src, threadIdx.x + half);
            fe_store<F>(tmp, threadIdx.x, fe_add<F>(y1, mr.times(fe_sub<F>(y2, y1))));
        }
        __syncthreads();
        src = tmp;
        len = half;
    }
}

// ---- reductions ------------------------------------------------------------------------------------
// Sums are accumulated LAZILY: a Wide is the plain integer sum (N + 1 limbs: up to 2^32 terms below 2^(32 N)), one
// carry chain per addition (N + 1 VALU ops instead of ~35 for a modular addition with its conditional subtraction),
// and is reduced mod p once per workgroup.  The result is the same fully reduced element (sum of the stored forms =
// stored form of the sum).
template <class F> struct Wide {
    uint32_t l[F::N + 1];
};
template <class F> __device__ __forceinline__ Wide<F> wide_zero() {
    Wide<F> w;
#pragma unroll
    for (int i = 0; i <= F::N; i++) w.l[i] = 0;
    return w;
}
template <class F> __device__ __forceinline__ void wide_add_fe(Wide<F> &w, const Fe<F> &a) {
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) w.l[i] = __builtin_addc(w.l[i], a.l[i], c, &c);
    w.l[F::N] += c;
}
template <class F> __device__ __forceinline__ void wide_add(Wide<F> &w, const Wide<F> &o) {
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i <= F::N; i++) w.l[i] = __builtin_addc(w.l[i], o.l[i], c, &c);
}
// S mod p.  Split S = lo + hi 2^s at s = bitlen(p) - 1, so lo < 2^s < p is already canonical and hi < 2^34 or so;
// hi 2^s mod p = mont(hi, 2^s R mod p) -- one product (on the latency path of every round), taken once per workgroup.
template <class F> __device__ __forceinline__ Fe<F> wide_reduce(const Wide<F> &w) {
    constexpr int N = F::N;
    const int tb = 31 - __builtin_clz(F::p(N - 1));       // bit s within the top limb (1..31 for all four moduli)
    Fe<F> lo, hi = fe_zero<F>(), k;
#pragma unroll
    for (int i = 0; i < N; i++) { lo.l[i] = w.l[i]; k.l[i] = F::red_k(i); }
    lo.l[N - 1] &= (1u << tb) - 1u;
    hi.l[0] = (w.l[N - 1] >> tb) | (w.l[N] << (32 - tb));
    hi.l[1] = w.l[N] >> tb;
    return fe_add<F>(lo, fe_mul<F>(hi, k));
}
// Cross-lane steps use DPP (VALU rate; __shfl_down would be a ds_bpermute per limb and makes a many-wave reduction
// LDS-crossbar bound).  Lanes without a source add zero.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
template <class F, int CTRL, int ROW_MASK> __device__ __forceinline__ void wide_dpp_step(Wide<F> &v) {
    Wide<F> o;
#pragma unroll
    for (int k = 0; k <= F::N; k++) o.l[k] = dpp_or_zero<CTRL, ROW_MASK>(v.l[k]);
    wide_add<F>(v, o);
}
// sum over each row of 16 lanes, valid in the row's lane 15 (row_shr:1,2,4,8)
template <class F> __device__ __forceinline__ void row_reduce_wide(Wide<F> &v) {
    wide_dpp_step<F, 0x111, 0xf>(v);
    wide_dpp_step<F, 0x112, 0xf>(v);
    wide_dpp_step<F, 0x114, 0xf>(v);
    wide_dpp_step<F, 0x118, 0xf>(v);
}
// sum over the wave, valid in lane 63 (then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).
// The first `active` of K independent sums are reduced (uniform), their instruction streams interleave.
template <class F, int K> __device__ __forceinline__ void wave_reduce_wide(Wide<F> (&v)[K], int active = K) {
#pragma unroll
    for (int k = 0; k < K; k++)
        if (k < active) {
            row_reduce_wide<F>(v[k]);
            wide_dpp_step<F, 0x142, 0xa>(v[k]);
            wide_dpp_step<F, 0x143, 0xc>(v[k]);
        }
}
// Workgroup sums of K lazily accumulated values.  `sh` holds K * (blockDim.x / 64) Wides.  Returns true in the K lanes
// (threads 0..K-1) that hold a result: thread k gets sum k, fully reduced.  Contains one __syncthreads; callers that
// reuse `sh` afterwards must synchronise again.
// `active_waves` (uniform, >= 1): only the first that many waves hold anything but zeros; the others skip their share of the work, which
// also frees issue slots on the SIMDs they share with the waves that matter (the one-workgroup tail of a sumcheck on short tables).
template <class F, int K> __device__ __forceinline__ bool block_reduce_wide(Wide<F> (&v)[K], Wide<F> *sh, Fe<F> &out, int active_waves = 0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int na = (active_waves > 0 && active_waves < nw) ? active_waves : nw;
    if (wave < na) {
        wave_reduce_wide<F, K>(v);
        if (lane == 63) {
#pragma unroll
            for (int k = 0; k < K; k++) sh[k * nw + wave] = v[k];
        }
    }
    __syncthreads();
    if ((int)threadIdx.x >= K) return false;
    Wide<F> tot = sh[threadIdx.x * nw];
    for (int w = 1; w < na; w++) wide_add<F>(tot, sh[threadIdx.x * nw + w]);
    out = wide_reduce<F>(tot);
    return true;
}

// partial sums of `nseg` equal contiguous segments of the table: partials[seg * gridDim.x + block]
// (nseg = 1: iter().sum() prover.rs:28 ; nseg = 2: split_polynomial_and_sum_each prover.rs:74-89)
template <class F> __global__ void segment_sums_kernel(const void *__restrict__ in, size_t seglen, int nseg,
                                                      void *__restrict__ partials) {
    __shared__ Wide<F> sh[kBlock / 64];
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int s = 0; s < nseg; s++) {
        Wide<F> acc[1] = {wide_zero<F>()};
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < seglen; i += stride)
            wide_add_fe<F>(acc[0], fe_load<F>(in, (size_t)s * seglen + i));
        Fe<F> tot;
        if (block_reduce_wide<F, 1>(acc, sh, tot)) fe_store<F>(partials, (size_t)s * gridDim.x + blockIdx.x, tot);
        __syncthreads();
    }
}
// one block: out[s] = sum of partials[s * count .. (s+1) * count)
template <class F> __global__ void finish_sums_kernel(const void *__restrict__ partials, size_t count, int nseg,
                                                     void *__restrict__ out) {
    __shared__ Wide<F> sh[kBlock / 64];
    for (int s = 0; s < nseg; s++) {
        Wide<F> acc[1] = {wide_zero<F>()};
        for (size_t i = threadIdx.x; i < count; i += blockDim.x) wide_add_fe<F>(acc[0], fe_load<F>(partials, (size_t)s * count + i));
        Fe<F> tot;
        if (block_reduce_wide<F, 1>(acc, sh, tot)) fe_store<F>(out, s, tot);
        __syncthreads();
    }
}

// ---- fused sumcheck round: fold variable 0 AND the folded table's two half sums ------------------------
// (prover.rs:50 of round k+1 fused into prover.rs:61-63 of round k.)  in has 4q elements,
// out has 2q; lane handles output indices i and i + q: 4 loads, 2 stores, 2 multiplications.
template <class F> __global__ void fold_half_sums_kernel(const void *__restrict__ in, void *__restrict__ out,
                                                        size_t q, Fe<F> r, void *__restrict__ partials,
                                                        const void *__restrict__ rp = nullptr) {
    __shared__ Wide<F> sh[2 * kBlock / 64];
    size_t stride = (size_t)gridDim.x * blockDim.x;
    Wide<F> sum[2] = {wide_zero<F>(), wide_zero<F>()};
    const Multiplier<F> mr(challenge_arg<F>(r, rp));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += stride) {
        Fe<F> a0 = fe_load<F>(in, i), a1 = fe_load<F>(in, i + q);
        Fe<F> b0 = fe_load<F>(in, i + 2 * q), b1 = fe_load<F>(in, i + 3 * q);
        Fe<F> o0 = fe_add<F>(a0, mr.times(fe_sub<F>(b0, a0)));
        Fe<F> o1 = fe_add<F>(a1, mr.times(fe_sub<F>(b1, a1)));
        fe_store<F>(out, i, o0);
        fe_store<F>(out, i + q, o1);
        wide_add_fe<F>(sum[0], o0);
        wide_add_fe<F>(sum[1], o1);
    }
    Fe<F> tot;
    if (block_reduce_wide<F, 2>(sum, sh, tot)) fe_store<F>(partials, (size_t)threadIdx.x * gridDim.x + blockIdx.x, tot);
}

// ---- several variables folded in one pass (basic sumcheck rounds: basic_multi.cuh; evaluate: zkmle_core.hip) ----------------------------
constexpr int kMultiMax = 8;            // variables per pass of the basic sumcheck (2^8 segment sums per exchange: basic_multi.cuh)
constexpr int kEvalMultiMax = 4;        // variables per pass of evaluate (zkmle_core.hip)
constexpr int kMultiBlocks = 1024;      // workgroups of a pass over a large table: (kMultiBlocks >> m) per segment of the output (r3 sweep: 1024-8192 within noise once the arrival atomics stopped sharing lines; 1024 = four workgroups per CU, one batch)

// out[j] = the table folded by r[0] (top variable), r[1], ... r[K-1], j < n = len >> K: a binary tree over in[j + i n], i < 2^K, whose
// level l pairs sub-trees 2^(K-1-l) entries apart (:61-63, K times).  Workgroups own one of the output's segments each (gridDim.x = nseg * bps)
// and leave its partial sums when `partials` is given.
// The exchange the LAST workgroup of a pass runs on the partials all its workgroups left (basic_multi.cuh; null counter: none).
struct HostMailbox;
struct MultiFin {
    unsigned *counter;           // the pass's arrival counter, then one per segment, 16 words apart; zero before the launch and left zero
    uint64_t *acc;               // 2^m x N words, kMultiAccStride words apart, zero before the launch and left zero: word (s, k) = the sum of limb k of segment s's partials
    int m;                       // 2^m segments, `bps` workgroups each
    uint64_t *limbs_out;         // sharded table: the 2^m sums leave as (N + 1) 32-bit limbs in 64-bit words for the all-reduce; nothing is posted
    HostMailbox *mb;
    uint64_t seq;
    void *proof;
    size_t chal_slot, per;       // challenge i goes to slot chal_slot + per i
    uint64_t *trace;             // measurement (ZK_PROOF_TRACE=1): wall_clock64 at [0] a workgroup's start, [1] the last arrival, [2] the post, [3] the answer
};
constexpr size_t kMultiAccStride = 16;                       // in 64-bit words: one 128-byte line per accumulator word
template <class F>
__device__ __forceinline__ void multi_finish_in_producer(const MultiFin &f, unsigned bps, const Fe<F> &tot);   // basic_multi.cuh

struct FoldKArgs {
    const void *in;
    void *out;
    size_t n;
    const void *r[kMultiMax];    // the challenges, on the device (proof slots)
    void *partials;              // nullptr: no sums (the tail takes over)
    unsigned bps;
};
// depth-first over the tree (the folds commute: the value is the multilinear extension at (r[0], ..)): K + 1 live values instead of 2^K
template <class F, int K, int L, int I> __device__ __forceinline__ Fe<F> fold_tree(const void *in, size_t j, size_t n, const Ufe<F> (&u)[K]) {
    if constexpr (L == K) {
        return fe_load<F>(in, j + (size_t)I * n);
    } else {
        const Fe<F> lo = fold_tree<F, K, L + 1, I>(in, j, n, u);
        const Fe<F> hi = fold_tree<F, K, L + 1, I + (1 << (K - 1 - L))>(in, j, n, u);
        return fe_add<F>(lo, fe_mul_u_pre<F>(u[L], fe_sub<F>(hi, lo)));
    }
}
// The same value as a weighted sum: out[j] = sum_i eq_i(r) in[j + i n], eq_i = prod_l (bit_{K-1-l}(i) ? r[l] : 1 - r[l]) -- the multilinear
// extension of the 2^K entries at (r[0], ..).  The tree costs 2^K - 1 full products (2 L^2 multiply-adds each); the weighted sum
// accumulates the 2^K raw integer products (L^2 each) in 64-bit columns and pays ONE Montgomery reduction (L^2) per output: 1377 instead of
// 2430 v_mad_u64_u32 for K = 4, which takes the kernel from the multiplier's roof back under the HBM roof.  Same field element (both are
// the canonical residue of the same value).
template <class F> struct RawAcc {
    uint64_t c[2 * UParams<F>::L];
};
// acc += x * w (integers; limbs of both below 2^29): every column gains at most L 2^58
template <class F> __device__ __forceinline__ void raw_mul_add(RawAcc<F> &acc, const Ufe<F> &x, const Ufe<F> &w) {
    constexpr int L = UParams<F>::L;
#pragma unroll
    for (int i = 0; i < L; i++) {
#pragma unroll
        for (int j = 0; j < L; j++) acc.c[i + j] += (uint64_t)x.l[j] * w.l[i];
    }
}
template <class F> __device__ __forceinline__ void raw_normalize(RawAcc<F> &acc) {
    constexpr int L = UParams<F>::L;
#pragma unroll
    for (int j = 0; j + 1 < 2 * L; j++) {
        acc.c[j + 1] += acc.c[j] >> UB;
        acc.c[j] &= UMASK;
    }
}
// acc / 2^(29 L) mod p for normalized columns: below acc / 2^(29 L) + p
template <class F> __device__ __forceinline__ Ufe<F> raw_mont_reduce(RawAcc<F> &acc) {
    constexpr int L = UParams<F>::L;
#pragma unroll
    for (int i = 0; i < L; i++) {
        const uint32_t m = ((uint32_t)acc.c[i] * UParams<F>::INV) & UMASK;
#pragma unroll
        for (int j = 0; j < L; j++) acc.c[i + j] += (uint64_t)m * UParams<F>::p(j);
        acc.c[i + 1] += acc.c[i] >> UB;                      // the low 29 bits of column i are now zero
    }
    Ufe<F> r;
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < L; j++) {
        const uint64_t v = acc.c[L + j] + c;
        r.l[j] = (uint32_t)v & UMASK;
        c = v >> UB;
    }
    return r;
}
constexpr int kRawCarryEvery = 4;       // products between two normalizations: 4 L 2^58 + 2^30 < 2^64 for L <= 14

// FIN: the pass's last workgroup runs the exchange on the partials (`fin`; translation units that include basic_multi.cuh only)
template <class F, int K, bool FIN = false> __global__ void __launch_bounds__(kBlock) foldk_seg_sums_kernel(FoldKArgs a, MultiFin fin) {
    __shared__ Wide<F> sh[kBlock / 64];
    const unsigned nseg = gridDim.x / a.bps, seg = blockIdx.x / a.bps, bq = blockIdx.x % a.bps;
    const size_t seglen = a.n / nseg, base = (size_t)seg * seglen, stride = (size_t)a.bps * blockDim.x;
    if (FIN && fin.trace && blockIdx.x == 0 && threadIdx.x == 0) fin.trace[0] = wall_clock64();
    Wide<F> acc[1] = {wide_zero<F>()};
    if constexpr (K <= 2) {
        Ufe<F> u[K];
#pragma unroll
        for (int k = 0; k < K; k++) u[k] = u_from_limbs32<F>(fe_load<F>(a.r[k], 0));
        for (size_t t = (size_t)bq * blockDim.x + threadIdx.x; t < seglen; t += stride) {
            const size_t j = base + t;
            const Fe<F> v = fold_tree<F, K, 0, 0>(a.in, j, a.n, u);
            fe_store<F>(a.out, j, v);
            wide_add_fe<F>(acc[0], v);
        }
    } else {
        __shared__ Ufe<F> sw[1 << K];                        // eq_i in the scan's form (x 2^(29 L)), fully reduced
        if (threadIdx.x < (1u << K)) {
            Fe<F> w = fe_one<F>();
#pragma unroll
            for (int l = 0; l < K; l++) {
                const Fe<F> r = fe_load<F>(a.r[l], 0);
                w = fe_mul<F>(w, ((threadIdx.x >> (K - 1 - l)) & 1u) ? r : fe_sub<F>(fe_one<F>(), r));
            }
            sw[threadIdx.x] = u_reduce_once<F>(u_from_std<F>(w));
        }
        __syncthreads();
        for (size_t t = (size_t)bq * blockDim.x + threadIdx.x; t < seglen; t += stride) {
            const size_t j = base + t;
            RawAcc<F> ra;
#pragma unroll
            for (int c = 0; c < 2 * UParams<F>::L; c++) ra.c[c] = 0;
            // (r3: fetching the next batch while this one is multiplied changed nothing -- 2^24 inputs x 81 multiply-adds each is what the
            // pass costs, ~105 us whatever K -- and took the kernel from 117 to 160 VGPRs)
#pragma unroll 1
            for (int i0 = 0; i0 < (1 << K); i0 += kRawCarryEvery) {
                Fe<F> x[kRawCarryEvery];
#pragma unroll
                for (int i = 0; i < kRawCarryEvery; i++) x[i] = fe_load<F>(a.in, j + (size_t)(i0 + i) * a.n);
#pragma unroll
                for (int i = 0; i < kRawCarryEvery; i++) raw_mul_add<F>(ra, u_from_limbs32<F>(x[i]), sw[i0 + i]);
                raw_normalize<F>(ra);
            }
            const Fe<F> v = u_to_limbs32<F>(u_reduce_once<F>(raw_mont_reduce<F>(ra)));
            fe_store<F>(a.out, j, v);
            wide_add_fe<F>(acc[0], v);
        }
    }
    if (a.partials == nullptr) return;
    Fe<F> tot;
    const bool have = block_reduce_wide<F, 1>(acc, sh, tot);
    if constexpr (FIN) {
        if (fin.counter) {
            if (threadIdx.x < 64) multi_finish_in_producer<F>(fin, a.bps, tot);
            return;
        }
    }
    if (have) fe_store<F>(a.partials, blockIdx.x, tot);
}

// The same pass with TWO lanes per output: a pass whose output is short for the chip (2^24 -> 2^17: 512 workgroups of one lane per output are two
// waves per SIMD, and the pass is as much multiply-adds as bytes: 106-108 us where its bytes alone are ~91) keeps four waves per SIMD this way.
// Lane part q sums the inputs i = q 2^(K-1) ..., the upper part hands its (reduced) half to the lower one through LDS.  Same weights, same exact sum.
// a.bps counts workgroups of kBlock / 2 outputs per segment; seglen is a multiple of a.bps * kBlock / 2 (fold_multi.h).
template <class F, int K, bool FIN = false> __global__ void __launch_bounds__(kBlock) foldk_seg_sums_split2_kernel(FoldKArgs a, MultiFin fin) {
    static_assert(K >= 4 && K <= 8, "2^(K-1) inputs per lane");
    constexpr int kOut = kBlock / 2, kPer = (1 << K) / 2;
    __shared__ Wide<F> sh[kBlock / 64];
    __shared__ Ufe<F> sw[1 << K];
    __shared__ Fe<F> parts[kOut];
    const unsigned nseg = gridDim.x / a.bps, seg = blockIdx.x / a.bps, bq = blockIdx.x % a.bps;
    const size_t seglen = a.n / nseg, base = (size_t)seg * seglen, stride = (size_t)a.bps * kOut;
    if (FIN && fin.trace && blockIdx.x == 0 && threadIdx.x == 0) fin.trace[0] = wall_clock64();
    if (threadIdx.x < (1u << K)) {
        Fe<F> w = fe_one<F>();
#pragma unroll
        for (int l = 0; l < K; l++) {
            const Fe<F> r = fe_load<F>(a.r[l], 0);
            w = fe_mul<F>(w, ((threadIdx.x >> (K - 1 - l)) & 1u) ? r : fe_sub<F>(fe_one<F>(), r));
        }
        sw[threadIdx.x] = u_reduce_once<F>(u_from_std<F>(w));
    }
    __syncthreads();
    const unsigned q = threadIdx.x / kOut, jj = threadIdx.x % kOut;
    Wide<F> acc[1] = {wide_zero<F>()};
    for (size_t t = (size_t)bq * kOut + jj; t < seglen; t += stride) {       // the same trip count for every lane of the workgroup
        const size_t j = base + t;
        RawAcc<F> ra;
#pragma unroll
        for (int c = 0; c < 2 * UParams<F>::L; c++) ra.c[c] = 0;
#pragma unroll 1
        for (int i0 = 0; i0 < kPer; i0 += kRawCarryEvery) {
            Fe<F> x[kRawCarryEvery];
#pragma unroll
            for (int i = 0; i < kRawCarryEvery; i++) x[i] = fe_load<F>(a.in, j + (size_t)(q * kPer + i0 + i) * a.n);
#pragma unroll
            for (int i = 0; i < kRawCarryEvery; i++) raw_mul_add<F>(ra, u_from_limbs32<F>(x[i]), sw[q * kPer + i0 + i]);
            raw_normalize<F>(ra);
        }
        Fe<F> v = u_to_limbs32<F>(u_reduce_once<F>(raw_mont_reduce<F>(ra)));
        if (q == 1) parts[jj] = v;
        __syncthreads();
        if (q == 0) {
            v = fe_add<F>(v, parts[jj]);
            fe_store<F>(a.out, j, v);
            wide_add_fe<F>(acc[0], v);
        }
        __syncthreads();
    }
    if (a.partials == nullptr) return;
    Fe<F> tot;
    const bool have = block_reduce_wide<F, 1>(acc, sh, tot);
    if constexpr (FIN) {
        if (fin.counter) {
            if (threadIdx.x < 64) multi_finish_in_producer<F>(fin, a.bps, tot);
            return;
        }
    }
    if (have) fe_store<F>(a.partials, blockIdx.x, tot);
}

// A pass with a SHORT output (n <= 2^13 entries: the last one before a one-workgroup tail) has too few outputs to fill the chip with one
// lane per output -- 2^11 outputs from 2^17 inputs are 8 workgroups whose lanes read 64 inputs one chunk after the other, 34 us for 4 MB
// (r3 trace).  Here kFoldSplit lanes share an output: lane part q sums inputs i = q * 2^K / kFoldSplit ... with the same weights and
// the same unreduced accumulation as foldk_seg_sums_kernel, reduces once, and the parts meet in LDS.  Field addition is exact, so the
// value is the one the sequential fold (mle.rs:118-127, K times) leaves.  No segment sums: the tail computes its own.
constexpr int kFoldSplit = 8;
constexpr int kFoldSplitBlock = 64;          // one wave per workgroup: 2^11 outputs are 256 workgroups, one per CU
template <class F, int K> __global__ void __launch_bounds__(kFoldSplitBlock) foldk_split_kernel(FoldKArgs a) {
    static_assert(K >= 4 && K <= 8, "2^K inputs per output, at least two per lane");
    constexpr int kOut = kFoldSplitBlock / kFoldSplit, kPer = (1 << K) / kFoldSplit, kStep = kPer < kRawCarryEvery ? kPer : kRawCarryEvery;
    __shared__ Ufe<F> sw[1 << K];
    __shared__ Fe<F> parts[kFoldSplit][kOut];
    for (unsigned e = threadIdx.x; e < (1u << K); e += kFoldSplitBlock) {
        Fe<F> w = fe_one<F>();
#pragma unroll
        for (int l = 0; l < K; l++) {
            const Fe<F> r = fe_load<F>(a.r[l], 0);
            w = fe_mul<F>(w, ((e >> (K - 1 - l)) & 1u) ? r : fe_sub<F>(fe_one<F>(), r));
        }
        sw[e] = u_reduce_once<F>(u_from_std<F>(w));
    }
    __syncthreads();
    const unsigned q = threadIdx.x / kOut, jj = threadIdx.x % kOut;
    const size_t j = (size_t)blockIdx.x * kOut + jj;
    if (j < a.n) {
        RawAcc<F> ra;
#pragma unroll
        for (int c = 0; c < 2 * UParams<F>::L; c++) ra.c[c] = 0;
#pragma unroll
        for (int i0 = 0; i0 < kPer; i0 += kStep) {
            Fe<F> x[kStep];
#pragma unroll
            for (int i = 0; i < kStep; i++) x[i] = fe_load<F>(a.in, j + (size_t)(q * kPer + i0 + i) * a.n);
#pragma unroll
            for (int i = 0; i < kStep; i++) raw_mul_add<F>(ra, u_from_limbs32<F>(x[i]), sw[q * kPer + i0 + i]);
            raw_normalize<F>(ra);
        }
        parts[q][jj] = u_to_limbs32<F>(u_reduce_once<F>(raw_mont_reduce<F>(ra)));
    }
    __syncthreads();
    if (q == 0 && j < a.n) {
        Fe<F> v = parts[0][jj];
#pragma unroll
        for (int p = 1; p < kFoldSplit; p++) v = fe_add<F>(v, parts[p][jj]);
        fe_store<F>(a.out, j, v);
    }
}

// compute_new_add_i_mul_i (gkr/src/utils.rs:23-68) in ONE pass: alpha fold(x, rb) + beta fold(x, rc), both chains folding the SAME k top variables of x, is
// sum_h (alpha eq(rb, h) + beta eq(rc, h)) x[h n + j] -- the reference's 2 k partial evaluations, two scalar products and one addition read x 2 (2 - 2^-k)
// times and launch 2 k + 3 kernels; this reads it once.  Field arithmetic is exact, so the entries are the same canonical elements.
constexpr int kFoldABMax = 8;
template <class F> struct FoldABArgs {
    Fe<F> rb[kFoldABMax], rc[kFoldABMax], alpha, beta;
    int k;
};
// `split` (a power of two, <= 2^k / 2 and <= kBlock / 16) lanes share an output: part q sums the inputs h = q 2^k / split ..., the parts meet in LDS (a short
// output -- 2^16 entries from a 2^23-entry predicate -- is 256 workgroups of one lane per output, each lane a chain of 128 dependent loads: 136 us for 268 MB).
// the 2^k weights alpha eq(rb, h) + beta eq(rc, h) in the scan's form, once per pass (2 k dependent products per weight: worked out in every workgroup
// of a 4096-workgroup pass they were 100 of its 160 us)
template <class F> __global__ void __launch_bounds__(1 << kFoldABMax) fold_alpha_beta_weights_kernel(FoldABArgs<F> a, Ufe<F> *__restrict__ w) {
    const unsigned e = threadIdx.x;
    if (e >= (1u << a.k)) return;
    Fe<F> wb = a.alpha, wc = a.beta;
    const Fe<F> one = fe_one<F>();
#pragma unroll 1
    for (int l = 0; l < a.k; l++) {                                  // variable 0 = the most significant bit (evaluation_form.rs:61-106 with var 0, k times)
        const bool bit = ((e >> (a.k - 1 - l)) & 1u) != 0;
        wb = fe_mul<F>(wb, bit ? a.rb[l] : fe_sub<F>(one, a.rb[l]));
        wc = fe_mul<F>(wc, bit ? a.rc[l] : fe_sub<F>(one, a.rc[l]));
    }
    w[e] = u_reduce_once<F>(u_from_std<F>(fe_add<F>(wb, wc)));
}
template <class F> __global__ void __launch_bounds__(kBlock) fold_alpha_beta_kernel(const void *__restrict__ in, void *__restrict__ out, size_t n, int k,
                                                                                    const Ufe<F> *__restrict__ w, unsigned split) {
    __shared__ Ufe<F> sw[1 << kFoldABMax];
    __shared__ Fe<F> parts[kBlock];
    const unsigned nw = 1u << k;
    for (unsigned e = threadIdx.x; e < nw; e += blockDim.x) sw[e] = w[e];
    __syncthreads();
    const unsigned kout = kBlock / split, per = nw / split;          // outputs per workgroup, inputs per lane (>= 2)
    const unsigned q = threadIdx.x / kout, jj = threadIdx.x % kout;
    const size_t nblk = (n + kout - 1) / kout;
    for (size_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {    // the same trip count for every lane of a workgroup
        const size_t j = blk * kout + jj;
        Fe<F> v = fe_zero<F>();
        if (j < n) {
            RawAcc<F> ra;
#pragma unroll
            for (int c = 0; c < 2 * UParams<F>::L; c++) ra.c[c] = 0;
            unsigned i0 = q * per;
            const unsigned i1 = i0 + per;
#pragma unroll 1
            for (; i0 + 4 <= i1; i0 += 4) {                          // four products between normalizations (kRawCarryEvery)
                Fe<F> x[4];
#pragma unroll
                for (int i = 0; i < 4; i++) x[i] = fe_load<F>(in, j + (size_t)(i0 + i) * n);
#pragma unroll
                for (int i = 0; i < 4; i++) raw_mul_add<F>(ra, u_from_limbs32<F>(x[i]), sw[i0 + i]);
                raw_normalize<F>(ra);
            }
            if (i0 < i1) {                                           // per = 2
                const Fe<F> x0 = fe_load<F>(in, j + (size_t)i0 * n), x1 = fe_load<F>(in, j + (size_t)(i0 + 1) * n);
                raw_mul_add<F>(ra, u_from_limbs32<F>(x0), sw[i0]);
                raw_mul_add<F>(ra, u_from_limbs32<F>(x1), sw[i0 + 1]);
                raw_normalize<F>(ra);
            }
            v = u_to_limbs32<F>(u_reduce_once<F>(raw_mont_reduce<F>(ra)));
        }
        if (split == 1) {
            if (j < n) fe_store<F>(out, j, v);
            continue;
        }
        parts[threadIdx.x] = v;
        __syncthreads();
        if (q == 0 && j < n) {
            for (unsigned s2 = 1; s2 < split; s2++) v = fe_add<F>(v, parts[s2 * kout + jj]);
            fe_store<F>(out, j, v);
        }
        __syncthreads();
    }
}

// ---- element-wise and tensor operations --------------------------------------------------------------
enum { OP_SCALAR_MUL = 0, OP_ADD = 1, OP_SUB_SCALAR = 2, OP_TO_CANONICAL_BE = 3, OP_HI_MINUS_LO = 4 };

template <class F, int OP> __global__ void elementwise_kernel(const void *__restrict__ a, const void *__restrict__ b,
                                                             void *__restrict__ out, size_t len, Fe<F> s) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) {
        Fe<F> x = fe_load<F>(a, i), o;
        if (OP == OP_SCALAR_MUL) o = fe_mul<F>(x, s);                         // evaluation_form.rs:53
        else if (OP == OP_ADD) o = fe_add<F>(x, fe_load<F>(b, i));            // :159
        else if (OP == OP_SUB_SCALAR) o = fe_sub<F>(x, s);                    // multilinear_kzg.rs:77
        else if (OP == OP_HI_MINUS_LO) o = fe_sub<F>(fe_load<F>(a, i + len), x);   // multilinear_kzg.rs:172-176
        else {                                                                // evaluation_form.rs:39
            Fe<F> c = fe_to_canonical<F>(x);                                  // into_bigint()
#pragma unroll
            for (int k = 0; k < F::N; k++) o.l[k] = __builtin_bswap32(c.l[F::N - 1 - k]);   // to_bytes_be()
        }
        fe_store<F>(out, i, o);
    }
}

// out[b * m + c] = wb[b] (+|*) wc[c]   evaluation_form.rs:116-120 / :136-140 (b-major)
template <class F, bool MUL> __global__ void tensor_kernel(const void *__restrict__ wb, const void *__restrict__ wc,
                                                          void *__restrict__ out, size_t m) {
    size_t total = m * m, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        Fe<F> x = fe_load<F>(wb, i / m), y = fe_load<F>(wc, i % m);
        fe_store<F>(out, i, MUL ? fe_mul<F>(x, y) : fe_add<F>(x, y));
    }
}

inline int grid_for(size_t work, int cap = kMaxBlocks) {
    size_t b = (work + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    if (b > (size_t)cap) b = cap;
    return (int)b;
}
// grid cap of the reduction kernels (one partial per block).  Measured r1, 2^24 fused round (fold_half_sums_kernel): with
// modular sums 2048 -> 188 us, 4096 -> 158, 16384 -> 170; with the lazy sums 2048 / 4096 / 8192 / 16384 -> 143 / 144 / 139 / 142
// (noise), while the one-workgroup finish kernel grows with the partial count (23 -> 34 us).  Whole provers, end of r1 (768 / 1536 /
// 3072 / 4096 / 6144 blocks): 2^24 basic rounds 0.871 / 0.878 / 0.883 / 0.896 / 0.892 ms, 4 x 2^22 GKR 1.072 / 1.071 / 1.078 / 1.093 / 1.117:
// 1536.  ZK_REDUCE_BLOCKS overrides.
inline int reduce_block_cap() {
    static const int v = [] {
        const char *e = getenv("ZK_REDUCE_BLOCKS");
        int k = e ? atoi(e) : 1536;
        return k < 64 ? 64 : (k > kMaxReduceBlocks ? kMaxReduceBlocks : k);
    }();
    return v;
}
inline int reduce_grid_for(size_t work) { return grid_for(work, reduce_block_cap()); }

}  // namespace zk
