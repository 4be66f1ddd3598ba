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

// kMultiMax (mle_kernels.cuh) = 8 rounds per exchange of a grid-wide pass: 2^8 segment sums go to the mailbox's `big` area, 8 challenges
// come back, one answer line each.  The one-workgroup tail runs up to six rounds per exchange (its sums and challenges live in LDS).
constexpr int kTailMultiMax = 6;

// ---- the exchange inside the producer ----------------------------------------------------------------------------------------
// r2 ran a one-workgroup kernel (multi_finish_kernel) behind every pass: reduce the partials, post, wait for the challenges.  r3: the
// pass's LAST workgroup to finish does it.  Every workgroup adds its (reduced) partial sum to its segment's accumulator -- N 64-bit words,
// word k = the sum of the partials' 32-bit limbs k: device-scope atomic adds, which meet at the memory side whatever XCD they come from --
// drains them and adds 1 to a device-scope counter; the workgroup whose add returns gridDim.x - 1 knows that every partial has landed,
// reads the 2^m x N words with device-scope loads, zeroes them for the next pass, propagates the carries and runs the exchange
// (MI355X_MICROARCH.md, inter-workgroup visibility: agent-scope atomics both sides; the last arriver learns it from the value its add
// returned; its other waves load behind a workgroup barrier).  No release / acquire fence: a release would write back the XCD's whole dirty
// L2 -- the pass's own output -- once per workgroup.  The first form of this (r3) kept one partial per workgroup and let the last
// workgroup gather the 2048 of them with sc1 loads: 6-15 us of dependent load rounds on the exchange's latency path.
// lane `lane` of wave 0 holds segment sums in registers and posts them itself: element `seg` of the mailbox's `big` area
template <class F> __device__ __forceinline__ void multi_post_element(HostMailbox *mb, unsigned seg, const Fe<F> &e) {
#pragma unroll
    for (int k = 0; k < F::N; k++) mb->big[seg * 12 + k] = e.l[k];
}
// after every lane's multi_post_element: publish request `seq`, wait for the m challenges, store them
template <class F> __device__ __forceinline__ void multi_publish_and_wait(const MultiFin &f, unsigned lane) {
    __threadfence_system();
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __atomic_store_n(&f.mb->gpu_seq, f.seq, __ATOMIC_RELEASE);
    const Fe<F> r = mailbox_wait_challenges<F>(f.mb, f.seq, lane, (unsigned)f.m);
    if ((int)lane < f.m) fe_store<F>(f.proof, f.chal_slot + f.per * lane, r);
}
// (N + 1) words holding sums of 32-bit limbs -> the field element
template <class F> __device__ __forceinline__ Fe<F> limb_words_reduce(const unsigned long long (&v)[F::N + 1]) {
    Wide<F> w;
    unsigned long long c = 0;
#pragma unroll
    for (int k = 0; k <= F::N; k++) {
        const unsigned long long x = v[k] + c;
        w.l[k] = (uint32_t)x;
        c = x >> 32;
    }
    return wide_reduce<F>(w);
}

// Wave 0 alone runs this (the workgroup's other waves have left: no workgroup barrier holds their slots while the atomics drain);
// `tot` is the workgroup's partial sum in lane 0 (block_reduce_wide's thread 0).  Everything stays in registers: lane l takes the
// segments l, l + 64, ... (up to 2^8 of them).
template <class F>
__device__ __forceinline__ void multi_finish_in_producer(const MultiFin &f, unsigned bps, const Fe<F> &tot) {
    const unsigned lane = threadIdx.x & 63u, nseg = 1u << f.m;
    unsigned last = 0;
    if (lane == 0) {
        // every accumulator word on a 128-byte line of its own: device-scope atomics to ONE line queue up behind each other at ~12 ns
        // apiece whatever their addresses (r3: with the 16 x 8 words packed into nine lines a pass of 2048 workgroups lost 13-20 us to its
        // 18 000 atomics, one of 8192 workgroups 100 us)
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(f.acc) + (size_t)(blockIdx.x / bps) * F::N * kMultiAccStride;
#pragma unroll
        for (int k = 0; k < F::N; k++)
            __hip_atomic_fetch_add(acc + (size_t)k * kMultiAccStride, (unsigned long long)tot.l[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // two levels of arrival counters, each on a line of its own: the `bps` workgroups of a segment meet at the segment's counter, the
        // last of each segment at the pass's.  ONE counter for all 2048 workgroups serialises them at ~11 ns per arrival.
        unsigned *segc = f.counter + 16u * (1u + blockIdx.x / bps);
        if (__hip_atomic_fetch_add(segc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == bps - 1u) {
            __hip_atomic_store(segc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           // left zero for the next launch (behind a kernel boundary)
            if (__hip_atomic_fetch_add(f.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nseg - 1u) {
                __hip_atomic_store(f.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1u;
            }
        }
    }
    if (!__builtin_amdgcn_readfirstlane(last)) return;
    if (f.trace && lane == 0) f.trace[1] = wall_clock64();
    for (unsigned seg = lane; seg < nseg; seg += 64) {
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(f.acc) + (size_t)seg * F::N * kMultiAccStride;
        unsigned long long v[F::N + 1];
#pragma unroll
        for (int k = 0; k < F::N; k++) v[k] = __hip_atomic_load(acc + (size_t)k * kMultiAccStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v[F::N] = 0;                                        // the top word only ever receives carries
#pragma unroll
        for (int k = 0; k < F::N; k++) __hip_atomic_store(acc + (size_t)k * kMultiAccStride, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (f.limbs_out) {                                   // sums of 32-bit limbs, carries unpropagated: what the all-reduce adds up
#pragma unroll
            for (int k = 0; k <= F::N; k++) f.limbs_out[(size_t)seg * (F::N + 1) + k] = v[k];
        } else {
            multi_post_element<F>(f.mb, seg, limb_words_reduce<F>(v));
        }
    }
    if (f.limbs_out) return;
    if (f.trace && lane == 0) f.trace[2] = wall_clock64();
    multi_publish_and_wait<F>(f, lane);
    if (f.trace && lane == 0) f.trace[3] = wall_clock64();
}

// partials[seg * bps + b] = sum over block b's share of segment seg (gridDim.x = nseg * bps); with `fin` the sums go to the pass's
// accumulators instead and the last workgroup runs the exchange
template <class F>
__global__ void __launch_bounds__(kBlock) seg_sums_kernel(const void *__restrict__ in, size_t seglen, unsigned bps, void *__restrict__ partials, MultiFin fin) {
    __shared__ Wide<F> sh[kBlock / 64];
    const unsigned seg = blockIdx.x / bps, bq = blockIdx.x % bps;
    const size_t base = (size_t)seg * seglen, stride = (size_t)bps * blockDim.x;
    if (fin.trace && blockIdx.x == 0 && threadIdx.x == 0) fin.trace[0] = wall_clock64();
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
    const bool have = block_reduce_wide<F, 1>(acc, sh, tot);
    if (fin.counter) {
        if (threadIdx.x < 64) multi_finish_in_producer<F>(fin, bps, tot);
    } else if (have) fe_store<F>(partials, blockIdx.x, tot);
}

// The exchange behind an all-reduce (sharded table): one wave takes the summed limb words of the 2^m segments, posts the sums and stores the
// m challenges the host answers with.
struct MultiArgs {
    const uint64_t *limbs_in;    // the sums over the ranks of the passes' limbs_out
    int m;
    HostMailbox *mb;
    uint64_t seq;
    void *proof;
    size_t chal_slot, per;       // challenge i goes to slot chal_slot + per i
};
template <class F> __global__ void __launch_bounds__(64) multi_finish_kernel(MultiArgs a) {
    const unsigned lane = threadIdx.x, nseg = 1u << a.m;
    for (unsigned seg = lane; seg < nseg; seg += 64) {
        unsigned long long v[F::N + 1];
#pragma unroll
        for (int k = 0; k <= F::N; k++) v[k] = a.limbs_in[(size_t)seg * (F::N + 1) + k];
        multi_post_element<F>(a.mb, seg, limb_words_reduce<F>(v));
    }
    const MultiFin f{nullptr, nullptr, a.m, nullptr, a.mb, a.seq, a.proof, a.chal_slot, a.per, nullptr};
    multi_publish_and_wait<F>(f, lane);
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
    __shared__ Fe<F> ev[1 << kTailMultiMax];
    __shared__ Fe<F> ch[kTailMultiMax];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const void *src = a.in;
    size_t cl = a.len, cs = a.chal_slot;
    uint64_t seq = a.seq0;
    while (cl >= 2) {
        const unsigned lg = 31u - (unsigned)__builtin_clz((unsigned)cl);
        const unsigned m = lg < (unsigned)kTailMultiMax ? lg : (unsigned)kTailMultiMax, nseg = 1u << m;
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
            mailbox_post<F>(a.mb, a.mb->big, ev, (int)nseg, seq, lane);
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
