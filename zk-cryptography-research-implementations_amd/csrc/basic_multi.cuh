// Basic sumcheck (prover.rs:35-71), several rounds per pass over the table -- host-assisted transcript step only (dev_transcript.cuh).
//
// A round of the basic sumcheck sends the two half sums of the current table (prover.rs:50, split_polynomial_and_sum_each :74-89) and
// folds the top variable by the challenge (:61-63).  Folding the top variable commutes with summing over the low bits: if S[0 .. 2^m) are
// the sums of the table over its 2^m contiguous segments (S is the table with its low variables summed out), then the segment sums of the
// folded table are the fold of S.  So the m rounds that follow are the basic sumcheck on the 2^m-entry table S, exactly, and need no pass
// over the big table: one kernel reduces 2^m segment sums, the host runs m transcript steps on them in ONE exchange (zkmle_sumcheck.hip
// serve_multi) and answers with the m challenges, and one kernel folds the m variables at once and leaves the segment sums of its output
// for the next exchange.  Per m rounds: (1 + 2^-m) table lengths of traffic instead of ~3, one exchange (and, sharded, one all-reduce of
// 2^m sums) instead of m.  Same field elements in the messages, same bytes absorbed, same challenges.
#pragma once
#include "dev_transcript.cuh"

namespace zk {

constexpr int kMultiMax = 4;            // rounds per exchange: 2^4 segment sums fit the mailbox's `fin` area, 4 challenges come back
constexpr int kMultiBlocks = 2048;      // workgroups of a pass over a large table: (kMultiBlocks >> m) per segment

// partials[seg * bps + b] = sum over block b's share of segment seg (gridDim.x = nseg * bps)
template <class F>
__global__ void __launch_bounds__(kBlock) seg_sums_kernel(const void *__restrict__ in, size_t seglen, unsigned bps, void *__restrict__ partials) {
    __shared__ Wide<F> sh[kBlock / 64];
    const unsigned seg = blockIdx.x / bps, bq = blockIdx.x % bps;
    const size_t base = (size_t)seg * seglen, stride = (size_t)bps * blockDim.x;
    Wide<F> acc[1] = {wide_zero<F>()};
    for (size_t t = (size_t)bq * blockDim.x + threadIdx.x; t < seglen; t += stride) wide_add_fe<F>(acc[0], fe_load<F>(in, base + t));
    Fe<F> tot;
    if (block_reduce_wide<F, 1>(acc, sh, tot)) fe_store<F>(partials, blockIdx.x, tot);
}

// out[j] = the table folded by r[0] (top variable), r[1], ... r[K-1], j < n = len >> K: a binary tree over in[j + i n], i < 2^K, whose
// level l pairs sub-trees 2^(K-1-l) entries apart (:61-63, K times).  Workgroups own one of the output's segments each (gridDim.x = nseg * bps)
// and leave its partial sums when `partials` is given.
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

template <class F, int K> __global__ void __launch_bounds__(kBlock) foldk_seg_sums_kernel(FoldKArgs a) {
    __shared__ Wide<F> sh[kBlock / 64];
    const unsigned nseg = gridDim.x / a.bps, seg = blockIdx.x / a.bps, bq = blockIdx.x % a.bps;
    const size_t seglen = a.n / nseg, base = (size_t)seg * seglen, stride = (size_t)a.bps * blockDim.x;
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
    if (block_reduce_wide<F, 1>(acc, sh, tot)) fe_store<F>(a.partials, blockIdx.x, tot);
}

// One workgroup of 2^m waves, wave w owns segment w: reduce its partials (or take the all-reduced limbs), then wave 0 posts the 2^m sums
// and stores the m challenges the host answers with.  With `limbs_out` the sums go out as (N + 1) 32-bit limbs in 64-bit words instead
// (the element-wise all-reduce over the ranks of a sharded table adds them without carries) and nothing is posted.
struct MultiArgs {
    const void *partials;
    size_t count;                // partials[seg * count + i]
    const uint64_t *limbs_in;    // non-null: the sums over the ranks of another launch's limbs_out
    uint64_t *limbs_out;
    int m;
    HostMailbox *mb;
    uint64_t seq;
    void *proof;
    size_t chal_slot, per;       // challenge i goes to slot chal_slot + per i
};
template <class F> __global__ void __launch_bounds__(64 << kMultiMax) multi_finish_kernel(MultiArgs a) {
    __shared__ Fe<F> ev[1 << kMultiMax];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (a.limbs_in) {
        if (lane == 0) {
            Wide<F> w;
            uint64_t c = 0;
#pragma unroll
            for (int k = 0; k <= F::N; k++) {                // words hold sums of 32-bit limbs: propagate the carries
                const uint64_t v = a.limbs_in[wave * (F::N + 1) + k] + c;
                w.l[k] = (uint32_t)v;
                c = v >> 32;
            }
            ev[wave] = wide_reduce<F>(w);
        }
    } else {
        Wide<F> acc[1] = {wide_zero<F>()};
        for (size_t i = lane; i < a.count; i += 64) wide_add_fe<F>(acc[0], fe_load<F>(a.partials, (size_t)wave * a.count + i));
        wave_reduce_wide<F, 1>(acc);
        if (lane == 63) {
            if (a.limbs_out) {
#pragma unroll
                for (int k = 0; k <= F::N; k++) a.limbs_out[wave * (F::N + 1) + k] = acc[0].l[k];
            } else {
                ev[wave] = wide_reduce<F>(acc[0]);
            }
        }
    }
    if (a.limbs_out) return;
    __syncthreads();
    if (wave != 0) return;
    mailbox_post<F>(a.mb, a.mb->fin, ev, 1 << a.m, a.seq, lane);
    mailbox_wait(a.mb, a.seq, lane);
    if ((int)lane < a.m) fe_store<F>(a.proof, a.chal_slot + a.per * lane, mailbox_element<F>(lane == 0 ? a.mb->chal : a.mb->aux[lane - 1]));
}

// Every round of a table of <= kTailLen entries (none of them started), one workgroup: per exchange up to kMultiMax rounds -- segment sums,
// post, the challenges, the folds level by level -- until 2^m <= 16 entries are left, which go to the host as they are (the "segment sums"
// of one-entry segments) and finish there.  The first level folds `in` into `buf` (the caller's table stays intact), the rest in place.
struct BasicTailArgs {
    const void *in;
    void *buf;                   // >= len / 2 entries
    size_t len;                  // 2 .. kTailLen, a power of two
    HostMailbox *mb;
    uint64_t seq0;               // request number of the first exchange
    void *proof;
    size_t chal_slot, per;       // of the tail's first round
};
template <class F> __global__ void __launch_bounds__(kTailBlock) basic_tail_kernel(BasicTailArgs a) {
    __shared__ Fe<F> ev[1 << kMultiMax];
    __shared__ Fe<F> ch[kMultiMax];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const void *src = a.in;
    size_t cl = a.len, cs = a.chal_slot;
    uint64_t seq = a.seq0;
    while (cl >= 2) {
        const unsigned lg = 31u - (unsigned)__builtin_clz((unsigned)cl);
        const unsigned m = lg < (unsigned)kMultiMax ? lg : (unsigned)kMultiMax, nseg = 1u << m;
        const size_t seglen = cl >> m;
        if (seglen == 1) {
            if (tid < nseg) ev[tid] = fe_load<F>(src, tid);
        } else {
            for (unsigned seg = wave; seg < nseg; seg += nwaves) {
                Wide<F> acc[1] = {wide_zero<F>()};
                for (size_t i = lane; i < seglen; i += 64) wide_add_fe<F>(acc[0], fe_load<F>(src, (size_t)seg * seglen + i));
                wave_reduce_wide<F, 1>(acc);
                if (lane == 63) ev[seg] = wide_reduce<F>(acc[0]);
            }
        }
        __syncthreads();
        if (wave == 0) {
            mailbox_post<F>(a.mb, a.mb->fin, ev, (int)nseg, seq, lane);
            if (seglen > 1) {
                mailbox_wait(a.mb, seq, lane);
                if (lane < m) {
                    const Fe<F> r = mailbox_element<F>(lane == 0 ? a.mb->chal : a.mb->aux[lane - 1]);
                    ch[lane] = r;
                    fe_store<F>(a.proof, cs + a.per * lane, r);
                }
            }
        }
        if (seglen == 1) break;                              // the host has the whole table: it runs the last rounds alone
        __syncthreads();
        for (unsigned l = 0; l < m; l++) {
            const size_t half = cl >> (l + 1);
            const Multiplier<F> mr(ch[l]);
            for (size_t i = tid; i < half; i += blockDim.x) {
                const Fe<F> x = fe_load<F>(src, i), y = fe_load<F>(src, i + half);
                fe_store<F>(a.buf, i, fe_add<F>(x, mr.times(fe_sub<F>(y, x))));
            }
            src = a.buf;
            __syncthreads();                                 // orders this level's global stores before the next level's loads
        }
        cl >>= m;
        cs += a.per * m;
        seq++;
    }
}

}  // namespace zk
