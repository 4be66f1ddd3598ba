// msm_reduce.hip -- Pippenger bucket combination and the log-depth weighted bucket reduction (own TU).
// Every array between the bucket kernel and the last kernel of the reduction holds XYZZ points in the internal 29-bit-limb
// form (g1u.cuh, 256 B per point): no conversion per product, which is what the late, single-lane levels are made of.
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "context.h"
#include "msm_kernels.cuh"
#include "g1u.cuh"
#include "msm_bits.cuh"

namespace zk {

// heavy buckets: sums of up to `group` consecutive partials of one bucket (one lane per output group)
__global__ void __launch_bounds__(256) msm_partials_regroup_kernel(const void *__restrict__ in_partials, const uint32_t *__restrict__ in_starts,
                                                                   const uint32_t *__restrict__ out_starts, size_t nbuckets, unsigned group,
                                                                   uint32_t nout, void *__restrict__ out_partials) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nout) return;
    size_t lo = 0, hi = nbuckets;
    while (hi - lo > 1) {
        size_t mid = (lo + hi) >> 1;
        if (out_starts[mid] <= t) lo = mid; else hi = mid;
    }
    size_t b = lo;
    uint32_t first = in_starts[b] + (t - out_starts[b]) * group;
    uint32_t end = in_starts[b + 1];
    if (first + group < end) end = first + group;
    G1XyzzU acc = g1u_load_xyzz(in_partials, first);
    for (uint32_t e = first + 1; e < end; e++) acc = g1u_add(acc, g1u_load_xyzz(in_partials, e));
    g1u_store_xyzz(out_partials, t, acc);
}

// bucket (w, b) = sum of its segments' partials, written to slot b of window w in the 2^(c-1)-slot reduction array A
// (slot b holds digit magnitude b + 1)
// (every slot is written, empty buckets as infinity; `B`, when given, receives the same values: the copy the two-stage reduction folds over l)
__global__ void __launch_bounds__(256) msm_bucket_combine_kernel(const void *__restrict__ partials, const uint32_t *__restrict__ seg_starts,
                                          unsigned nwin, unsigned c, void *__restrict__ A, void *__restrict__ B) {
    unsigned nb = 1u << (c - 1);
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)nwin * nb) return;
    unsigned w = id / nb, b = id % nb;
    uint32_t s0 = seg_starts[id], s1 = seg_starts[id + 1];
    G1XyzzU acc = g1u_inf();
    for (uint32_t s = s0; s < s1; s++) acc = g1u_add(acc, g1u_load_xyzz(partials, s));
    g1u_store_xyzz(A, ((size_t)w << (c - 1)) + b, acc);
    if (B) g1u_store_xyzz(B, ((size_t)w << (c - 1)) + b, acc);
}

// step 4: one halving level of  sum_b b A[b]  over the 2^(c-1) slots of every window, in place (half = current length / 2):
// A' = A_lo + A_hi, R' = A_hi + 2 (R_lo + R_hi), ending with R[0] = sum_b b A[b] and A[0] = sum_b A[b].  The slot weights are
// b + 1, so the window sum is R[0] + A[0] (msm_window_sums_kernel): one level and half the array less than reducing 2^c slots
// indexed by the digit magnitude itself (r1: 1.75 -> see DESIGN.md).
// The two updates of a pair are independent (A' writes the low half of A, R' reads the high half of A), so they run in
// different lanes: the first nwin * half lanes take A', the next nwin * half take R' -- the level's critical path is
// add, double, add instead of those plus one more add (the late levels are pure latency).
// QUAD: four lanes per work item (g1u_add_quad / g1u_dbl_quad) for the short, latency-bound levels.
template <bool QUAD>
__global__ void __launch_bounds__(256) msm_reduce_level_kernel(void *__restrict__ A, void *__restrict__ R, unsigned nwin, unsigned c, size_t half) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned q = QUAD ? (unsigned)(id & 3u) : 0u;
    if (QUAD) id >>= 2;
    const size_t work = (size_t)nwin * half;
    if (id >= 2 * work) return;
    const bool second = id >= work;
    if (second) id -= work;
    size_t w = id / half, b = id % half;
    size_t base = w << (c - 1);
    G1XyzzU ahi = g1u_load_xyzz(A, base + b + half);
    if (!second) {
        G1XyzzU alo = g1u_load_xyzz(A, base + b);
        G1XyzzU r = QUAD ? g1u_add_quad(alo, ahi, q) : g1u_add(alo, ahi);
        if (q == 0) g1u_store_xyzz(A, base + b, r);
    } else {
        G1XyzzU rlo = g1u_load_xyzz(R, base + b), rhi = g1u_load_xyzz(R, base + b + half);
        G1XyzzU r = QUAD ? g1u_add_quad(ahi, g1u_dbl_quad(g1u_add_quad(rlo, rhi, q), q), q) : g1u_add(ahi, g1u_dbl(g1u_add(rlo, rhi)));
        if (q == 0) g1u_store_xyzz(R, base + b, r);
    }
}


// ---- two-stage form of the weighted bucket sum -----------------------------------------------------------------------------------
// sum_b (b + 1) A[b] over nb = H * L slots, b = h L + l:   = L * sum_h h D[h]  +  sum_l (l + 1) C[l],
// with the plain sums C[l] = sum_h A[h][l] and D[h] = sum_l A[h][l].  The plain sums are halving trees of ONE addition per
// level (A reduces over h in place, a copy of it reduces over l), and only the two short arrays C and D go through the
// 3-operation weighted levels: log2(nb) levels of (add, double, add) become log2(H) plain + log2(L) weighted levels.
template <bool QUAD>
__global__ void __launch_bounds__(256) msm_plain_level_kernel(void *__restrict__ A, void *__restrict__ B, unsigned nwin, unsigned cm1, unsigned k,
                                                              size_t hh, size_t lh) {
    // role 1: A[w][h][l] += A[w][h + hh][l], h < hh (all l);  role 2: B[w][h][l] += B[w][h][l + lh], l < lh (all h)
    const size_t L = (size_t)1 << k, H = (size_t)1 << (cm1 - k);
    const size_t work1 = (size_t)nwin * hh * L, work2 = (size_t)nwin * H * lh;
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned q = QUAD ? (unsigned)(id & 3u) : 0u;
    if (QUAD) id >>= 2;
    if (id < work1) {
        size_t w = id / (hh * L), r = id % (hh * L);
        size_t base = (w << cm1) + r;
        G1XyzzU x = g1u_load_xyzz(A, base), y = g1u_load_xyzz(A, base + hh * L);
        G1XyzzU s = QUAD ? g1u_add_quad(x, y, q) : g1u_add(x, y);
        if (q == 0) g1u_store_xyzz(A, base, s);
    } else if (id < work1 + work2) {
        id -= work1;
        size_t w = id / (H * lh), r = id % (H * lh);
        size_t h = r / lh, l = r % lh;
        size_t base = (w << cm1) + h * L + l;
        G1XyzzU x = g1u_load_xyzz(B, base), y = g1u_load_xyzz(B, base + lh);
        G1XyzzU s = QUAD ? g1u_add_quad(x, y, q) : g1u_add(x, y);
        if (q == 0) g1u_store_xyzz(B, base, s);
    }
}
// compact the two short arrays, zero-padded to M = 2^mbits entries: X[0][w][l] = C[l] = A[w][l],  X[1][w][h] = D[h] = B[w][h L]
__global__ void msm_gather_cd_kernel(const void *__restrict__ A, const void *__restrict__ B, unsigned nwin, unsigned cm1, unsigned k, unsigned mbits,
                                     void *__restrict__ X) {
    const size_t L = (size_t)1 << k, H = (size_t)1 << (cm1 - k), M = (size_t)1 << mbits;
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 2 * (size_t)nwin * M) return;
    size_t p = id / ((size_t)nwin * M), r = id % ((size_t)nwin * M), w = r / M, i = r % M;
    G1XyzzU v = g1u_inf();
    if (p == 0) { if (i < L) v = g1u_load_xyzz(A, (w << cm1) + i); }
    else if (i < H) v = g1u_load_xyzz(B, (w << cm1) + i * L);
    g1u_store_xyzz(X, id, v);
}
// out[3 w .. 3 w + 2] = sum_l l C[l], sum_l C[l], sum_h h D[h]  (R_l[0], A_l[0], R_h[0] of the weighted reductions).
// The results leave the internal form here (stored XYZZ, canonical limbs) for the host's window combination: one lane per
// coordinate (12 per window), one product each.
__global__ void msm_two_stage_out_kernel(const void *__restrict__ X, const void *__restrict__ Y, unsigned nwin, unsigned mbits, void *__restrict__ out) {
    unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 12u * nwin) return;
    const unsigned w = id / 12u, pt = (id % 12u) / 4u, co = id % 4u;
    const size_t M = (size_t)1 << mbits;
    const void *src = pt == 1 ? X : Y;
    const size_t idx = pt == 2 ? ((size_t)nwin + w) * M : (size_t)w * M;
    const uint32_t *p = reinterpret_cast<const uint32_t *>(src) + idx * (4 * kUWords);
    const bool inf = u_is_exact_zero<Fq381>(fqu_load(p + 2 * kUWords));
    fe_store<Fq>(out, 4 * (3 * (size_t)w + pt) + co, inf ? fe_zero<Fq>() : u_to_std<Fq381>(fqu_load(p + co * kUWords)));
}
// every level of the weighted reduction of the short arrays (X: values, Y: running weighted sums, M = 2^mbits entries each) in
// ONE launch: a workgroup per array, a quad per item, __syncthreads between the levels (level after level as separate launches
// cost ~8 us of launch, load and store latency each, on top of the arithmetic).  Same operations as msm_reduce_level_kernel.
__global__ void __launch_bounds__(512) msm_weighted_tail_kernel(void *__restrict__ X, void *__restrict__ Y, unsigned mbits, size_t first_half) {
    const size_t base = (size_t)blockIdx.x << mbits;
    const unsigned quad = threadIdx.x >> 2, q = threadIdx.x & 3u, nquads = blockDim.x >> 2;
    for (size_t half = first_half; half >= 1; half >>= 1) {
        for (size_t item = quad; item < 2 * half; item += nquads) {
            const bool second = item >= half;
            const size_t b = second ? item - half : item;
            G1XyzzU ahi = g1u_load_xyzz(X, base + b + half);
            if (!second) {
                G1XyzzU r = g1u_add_quad(g1u_load_xyzz(X, base + b), ahi, q);
                if (q == 0) g1u_store_xyzz(X, base + b, r);
            } else {
                G1XyzzU rlo = g1u_load_xyzz(Y, base + b), rhi = g1u_load_xyzz(Y, base + b + half);
                G1XyzzU r = g1u_add_quad(ahi, g1u_dbl_quad(g1u_add_quad(rlo, rhi, q), q), q);
                if (q == 0) g1u_store_xyzz(Y, base + b, r);
            }
        }
        __syncthreads();                                     // workgroup-scope: the level's stores are visible to the next level
    }
}

__global__ void msm_window_sums_kernel(const void *__restrict__ A, const void *__restrict__ R, unsigned nwin, unsigned c, void *__restrict__ out) {
    unsigned w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwin) return;
    size_t base = (size_t)w << (c - 1);
    g1_store_xyzz(out, w, g1u_to_std(g1u_add(g1u_load_xyzz(R, base), g1u_load_xyzz(A, base))));
}

int launch_msm_partials_regroup(const void *in_partials, const uint32_t *in_starts, const uint32_t *out_starts, size_t nbuckets,
                                unsigned group, uint32_t nout, void *out_partials, hipStream_t s) {
    msm_partials_regroup_kernel<<<(nout + 255) / 256, 256, 0, s>>>(in_partials, in_starts, out_starts, nbuckets, group, nout, out_partials);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int launch_msm_bucket_combine(const void *partials, const uint32_t *seg_starts, unsigned nwin, unsigned c, void *A, void *B, hipStream_t s) {
    size_t nbuckets = (size_t)nwin << (c - 1);
    msm_bucket_combine_kernel<<<(unsigned)((nbuckets + 255) / 256), 256, 0, s>>>(partials, seg_starts, nwin, c, A, B);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int launch_msm_window_sums(const void *A, const void *R, unsigned nwin, unsigned c, void *out, hipStream_t s) {
    msm_window_sums_kernel<<<(nwin + 63) / 64, 64, 0, s>>>(A, R, nwin, c, out);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
// levels with at most this many additions use four lanes per addition (measured: a level of <= ~2^13 additions is latency)
constexpr size_t kQuadLevelWork = (size_t)1 << 13;
int launch_msm_plain_level(void *A, void *B, unsigned nwin, unsigned cm1, unsigned k, size_t hh, size_t lh, hipStream_t s) {
    size_t work = (size_t)nwin * ((hh << k) + (((size_t)1 << (cm1 - k)) * lh));
    if (work == 0) return ZK_OK;
    if (work <= kQuadLevelWork) msm_plain_level_kernel<true><<<(unsigned)((4 * work + 255) / 256), 256, 0, s>>>(A, B, nwin, cm1, k, hh, lh);
    else msm_plain_level_kernel<false><<<(unsigned)((work + 255) / 256), 256, 0, s>>>(A, B, nwin, cm1, k, hh, lh);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int launch_msm_gather_cd(const void *A, const void *B, unsigned nwin, unsigned cm1, unsigned k, unsigned mbits, void *X, hipStream_t s) {
    size_t work = 2 * ((size_t)nwin << mbits);
    msm_gather_cd_kernel<<<(unsigned)((work + 255) / 256), 256, 0, s>>>(A, B, nwin, cm1, k, mbits, X);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int launch_msm_two_stage_out(const void *X, const void *Y, unsigned nwin, unsigned mbits, void *out, hipStream_t s) {
    msm_two_stage_out_kernel<<<(12 * nwin + 63) / 64, 64, 0, s>>>(X, Y, nwin, mbits, out);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
// the weighted sums of `narrays` arrays of 2^mbits entries by bits (mbits <= 8: a workgroup per (array, bit) holds its 2^(mbits-1) entries in
// at most two rounds of 64 quads); S: narrays x (mbits + 1) points of scratch
int launch_msm_weighted_bits(void *X, void *Y, unsigned narrays, unsigned mbits, void *S, hipStream_t s) {
    if (mbits < 1 || mbits > 8) return ZK_E_ARG;
    msm_bit_sums_kernel<true><<<dim3(narrays, mbits + 1), 4 * kBitQuads, 0, s>>>(X, mbits, S);
    msm_bit_combine_kernel<3><<<narrays, 64, 0, s>>>(S, mbits, X, Y);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int launch_msm_reduce_level(void *A, void *R, unsigned nwin, unsigned c, size_t half, hipStream_t s);
int launch_msm_weighted_tail(void *X, void *Y, unsigned narrays, unsigned mbits, hipStream_t s) {
    // a level of `half` pairs is 2 * half additions per array: the one-workgroup tail has 128 quads, so levels wider than that are
    // throughput of ONE CU there (r3: 1.35 ms for two arrays of 2^11 entries) and go grid-wide instead, one launch each
    size_t half = (size_t)1 << (mbits - 1);
    for (; half > 64; half >>= 1) ZK_TRY(launch_msm_reduce_level(X, Y, narrays, mbits + 1, half, s));
    msm_weighted_tail_kernel<<<narrays, 512, 0, s>>>(X, Y, mbits, half);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int launch_msm_reduce_level(void *A, void *R, unsigned nwin, unsigned c, size_t half, hipStream_t s) {
    size_t work = 2 * (size_t)nwin * half;
    if (work <= kQuadLevelWork) msm_reduce_level_kernel<true><<<(unsigned)((4 * work + 255) / 256), 256, 0, s>>>(A, R, nwin, c, half);
    else msm_reduce_level_kernel<false><<<(unsigned)((work + 255) / 256), 256, 0, s>>>(A, R, nwin, c, half);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

}  // namespace zk
