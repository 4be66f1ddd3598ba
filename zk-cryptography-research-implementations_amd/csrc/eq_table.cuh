// eq_table.cuh -- the eq / Lagrange table  L_idx(tau) = prod_i (bit_i(idx) ? tau_i : 1 - tau_i), variable 0 = MSB
// (compute_lagrange_basis trusted_setup.rs:24-49; the same table is the eq(point, .) of the sparse GKR prover).
//
// The reference doubles the table once per variable.  Level by level that is one launch per variable, and the
// first dozen launches are latency only (measured: 22 launches x ~7.6 us for a 2^22 table).  Here the table is the
// outer product of the tables of its high and low variable halves, recursively, so a 2^22 table is 4 tiny direct
// kernels (<= 6 variables: the product is taken directly, 6 multiplications per entry) and 3 outer-product
// kernels, the last of which does all the real work: one multiplication and one 32-byte store per entry.
// Products of the same factors in another order are the same field element: the table is unchanged bit for bit.
#pragma once
#include <vector>

#include "context.h"
#include "mle_kernels.cuh"

namespace zk {

constexpr int kEqDirectBits = 6;

template <class F> struct EqPoint {
    Fe<F> tau[kEqDirectBits];
};

// out[idx] = prod_i (bit ? tau_i : 1 - tau_i) over nbits <= kEqDirectBits variables, MSB first
template <class F> __global__ void eq_direct_kernel(EqPoint<F> pt, int nbits, void *__restrict__ out) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (1u << nbits)) return;
    Fe<F> acc = fe_one<F>();
#pragma unroll 1
    for (int i = 0; i < nbits; i++) {
        const Fe<F> t = pt.tau[i];
        const bool bit = (idx >> (nbits - 1 - i)) & 1u;
        acc = fe_mul<F>(acc, bit ? t : fe_sub<F>(fe_one<F>(), t));
    }
    fe_store<F>(out, idx, acc);
}

// out[i] = hi[i >> lbits] * lo[i & (2^lbits - 1)]
template <class F> __global__ void eq_outer_kernel(const void *__restrict__ hi, const void *__restrict__ lo, unsigned lbits, size_t n,
                                                   void *__restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, mask = ((size_t)1 << lbits) - 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        fe_store<F>(out, i, fe_mul<F>(fe_load<F>(hi, i >> lbits), fe_load<F>(lo, i & mask)));
}

// Enqueue the construction of the 2^nbits-entry table of `point` (nbits elements, u64 limbs) into `out` (device).
// Temporaries come from the caching pool and are returned to it when the caller calls release() -- after the
// stream has been synchronised or further work on the same stream has been enqueued (the pool is stream-ordered by
// construction: every kernel of this library runs on the null stream).
template <class F> struct EqBuilder {
    std::vector<void *> temps;
    ~EqBuilder() { release(); }
    void release() {
        for (void *p : temps) pool_free(p);
        temps.clear();
    }
    int build(const uint64_t *point, uint32_t nbits, void *out) {
        if (nbits <= (uint32_t)kEqDirectBits) {
            EqPoint<F> pt;
            for (int i = 0; i < kEqDirectBits; i++) {
                if ((uint32_t)i < nbits) memcpy(pt.tau[i].l, point + (size_t)i * (F::N / 2), 4 * F::N);
                else pt.tau[i] = fe_zero<F>();
            }
            eq_direct_kernel<F><<<1, 64, 0, cur_stream()>>>(pt, (int)nbits, out);
            ZK_HIP(hipGetLastError());
            return ZK_OK;
        }
        const uint32_t hbits = nbits / 2, lbits = nbits - hbits;       // variables 0..hbits-1 are the high index bits
        void *hi = nullptr, *lo = nullptr;
        ZK_TRY(pool_alloc(((size_t)1 << hbits) * 4 * F::N, &hi));
        temps.push_back(hi);
        ZK_TRY(pool_alloc(((size_t)1 << lbits) * 4 * F::N, &lo));
        temps.push_back(lo);
        ZK_TRY(build(point, hbits, hi));
        ZK_TRY(build(point + (size_t)hbits * (F::N / 2), lbits, lo));
        const size_t n = (size_t)1 << nbits;
        eq_outer_kernel<F><<<grid_for(n), kBlock, 0, cur_stream()>>>(hi, lo, lbits, n, out);
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }
};

}  // namespace zk
