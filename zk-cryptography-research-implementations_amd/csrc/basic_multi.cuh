// Basic sumcheck (prover.rs:35-71), several rounds per pass over the table -- host-assisted transcript step only (dev_transcript.cuh).
//
// A round of the basic sumcheck sends the two half sums of the current table (prover.rs:50, split_polynomial_and_sum_each :74-89) and
// folds the top variable by the challenge (:61-63).  Folding the top variable commutes with summing over the low bits: if S[0 .. 2^m) are
// the sums of the table over its 2^m contiguous segments (S is the table with its low variables summed out), then the segment sums of the
// folded table are the fold of S.  So the m rounds that follow are the basic sumcheck on the 2^m-entry table S, exactly, and need no pass
// over the big table: one kernel reduces 2^m segment sums, the host runs m transcript steps on them in ONE exchange (zkmle_sumcheck.hip
// serve_multi) and answers with the m challenges, and one kernel folds the m variables at once and leaves the segment sums of its output
// (foldk_seg_sums_kernel, mle_kernels.cuh) for the next exchange.  Per m rounds: (1 + 2^-m) table lengths of traffic instead of ~3, one exchange (and, sharded, one all-reduce of
// 2^m sums) instead of m.  Same field elements in the messages, same bytes absorbed, same challenges.
#pragma once
#include "dev_transcript.cuh"

namespace zk {

// kMultiMax (mle_kernels.cuh) rounds per exchange: 2^4 segment sums fit the mailbox's `fin` area, 4 challenges come back

// partials[seg * bps + b] = sum over block b's share of segment seg (gridDim.x = nseg * bps)
template <class F>
__global__ void __launch_bounds__(kBlock) seg_sums_kernel(const void *__restrict__ in, size_t seglen, unsigned bps, void *__restrict__ partials) {
    __shared__ Wide<F> sh[kBlock / 64];
    const unsigned seg = blockIdx.x / bps, bq = blockIdx.x % bps;
    const size_t base = (size_t)seg * seglen, stride = (size_t)bps * blockDim.x;
    Wide<F> acc[1] = {wide_zero<F>()};
    size_t t = (size_t)bq * blockDim.x + threadIdx.x;
    for (; t + 3 * stride < seglen; t += 4 * stride) {       // four loads in flight per lane
        const Fe<F> x0 = fe_load<F>(in, base + t), x1 = fe_load<F>(in, base + t + stride);
        const Fe<F> x2 = fe_load<F>(in, base + t + 2 * stride), x3 = fe_load<F>(in, base + t + 3 * stride);
        wide_add_fe<F>(acc[0], x0);
        wide_add_fe<F>(acc[0], x1);
        wide_add_fe<F>(acc[0], x2);
        wide_add_fe<F>(acc[0], x3);
    }
    for (; t < seglen; t += stride) wide_add_fe<F>(acc[0], fe_load<F>(in, base + t));
    Fe<F> tot;
    if (block_reduce_wide<F, 1>(acc, sh, tot)) fe_store<F>(partials, blockIdx.x, tot);
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
    const Fe<F> r = mailbox_wait_challenges<F>(a.mb, a.seq, lane, (unsigned)a.m);
    if ((int)lane < a.m) fe_store<F>(a.proof, a.chal_slot + a.per * lane, r);
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
                const Fe<F> r = mailbox_wait_challenges<F>(a.mb, seq, lane, m);
                if (lane < m) {
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
