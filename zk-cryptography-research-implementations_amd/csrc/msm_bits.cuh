// msm_bits.cuh -- the weighted sums of the bucket reduction's short arrays by the bits of the weight (included by msm_reduce.hip and by
// tools/test_quad_ops.hip, which runs these kernels on multiples of the generator against a one-lane Horner sum).
#pragma once
#include "msm_kernels.cuh"
#include "g1u.cuh"

namespace zk {

// ---- the weighted sums of the short arrays by the BITS of the weight (r4) ---------------------------------------------------------
// sum_i i X[i] = sum_j 2^j S_j with S_j = the plain sum of the entries whose index has bit j set.  The halving levels above pay three DEPENDENT
// point operations per level (add, double, add) and a level of a 2^8-entry array is pure latency (~50 us each on quads: 353 us for the seven
// levels of config 3's tail, r3 timeline).  Here every S_j is a tree of ONE addition per level, all of them side by side (a workgroup per
// (array, bit): 2 nwin (mbits + 1) workgroups), and the 2^j are applied afterwards on mbits values: log2 levels of (w doublings, one addition).
// Depth for 2^8 entries: 7 additions + 7 doublings + 3 additions, where the levels took 8 x 3 operations.  Same group element.
// X: narrays arrays of M = 2^mbits entries.  S[a * (mbits + 1) + j] = S_j of array a, j < mbits; [.. + mbits] = the plain total.
constexpr int kBitQuads = 64;
template <bool QUAD> __global__ void __launch_bounds__(4 * kBitQuads) msm_bit_sums_kernel(const void *__restrict__ X, unsigned mbits, void *__restrict__ S) {
    __shared__ uint4 lds[kBitQuads * (kXyzzUBytes / 16)];
    const unsigned a = blockIdx.x, j = blockIdx.y, quad = threadIdx.x >> 2, q = threadIdx.x & 3u;
    const size_t base = (size_t)a << mbits;
    const unsigned cnt = j == mbits ? 1u << mbits : 1u << (mbits - 1);          // entries this workgroup adds up
    G1XyzzU v = g1u_inf();
    for (unsigned t = quad; t < cnt; t += kBitQuads) {                           // entry t of the selection: bit j of its index is set
        const size_t i = j == mbits ? t : ((((size_t)t >> j) << (j + 1)) | ((size_t)1 << j) | (t & ((1u << j) - 1u)));
        v = QUAD ? g1u_add_quad(v, g1u_load_xyzz(X, base + i), q) : g1u_add(v, g1u_load_xyzz(X, base + i));
    }
    for (unsigned s = kBitQuads / 2; s >= 1; s >>= 1) {                          // tree over the quads' sums, through LDS
        if (quad >= s && quad < 2 * s && q == 0) g1u_store_xyzz(lds, quad - s, v);
        __syncthreads();
        if (quad < s) v = QUAD ? g1u_add_quad(v, g1u_load_xyzz(lds, quad), q) : g1u_add(v, g1u_load_xyzz(lds, quad));
        __syncthreads();
    }
    if (quad == 0 && q == 0) g1u_store_xyzz(S, (size_t)a * (mbits + 1) + j, v);
}
// weighted[a] = sum_j 2^j S[a][j] -> Y[a M]; total[a] -> X[a M] (where msm_two_stage_out_kernel reads them).  One workgroup per array, a quad per
// bit; pairs of groups merge as  G_g + 2^w G_(g + stride): w doublings on the odd group, one addition, w and stride double per step.
template <int QUAD> __global__ void __launch_bounds__(64) msm_bit_combine_kernel(const void *__restrict__ S, unsigned mbits, void *__restrict__ X, void *__restrict__ Y) {
    __shared__ uint4 lds[16 * (kXyzzUBytes / 16)];
    const unsigned a = blockIdx.x, g = threadIdx.x >> 2, q = threadIdx.x & 3u;
    unsigned ng = 1;
    while (ng < mbits) ng <<= 1;                                                 // mbits <= 16: one quad per group
    G1XyzzU v = g < mbits ? g1u_load_xyzz(S, (size_t)a * (mbits + 1) + g) : g1u_inf();
    for (unsigned stride = 1, w = 1; stride < ng; stride <<= 1, w <<= 1) {
        // every quad runs the same instructions (the quad operations move data between lanes: no divergence around them); who keeps what is a select
        G1XyzzU dv = v;
        for (unsigned d = 0; d < w; d++) dv = (QUAD & 1) ? g1u_dbl_quad(dv, q) : g1u_dbl(dv);
        if ((g & (2 * stride - 1)) == stride && q == 0) g1u_store_xyzz(lds, g, dv);
        __syncthreads();
        const G1XyzzU other = g1u_load_xyzz(lds, (g + stride) & 15u);
        const G1XyzzU sum = (QUAD & 2) ? g1u_add_quad(v, other, q) : g1u_add(v, other);
        if ((g & (2 * stride - 1)) == 0 && g + stride < ng) v = sum;
        __syncthreads();
    }
    if (g == 0 && q == 0) {
        g1u_store_xyzz(Y, (size_t)a << mbits, v);
        g1u_store_xyzz(X, (size_t)a << mbits, g1u_load_xyzz(S, (size_t)a * (mbits + 1) + mbits));
    }
}

}  // namespace zk
