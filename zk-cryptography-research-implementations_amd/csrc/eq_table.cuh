// eq_table.cuh -- the eq / Lagrange table  L_idx(tau) = prod_i (bit_i(idx) ? tau_i : 1 - tau_i), variable 0 = MSB
// (compute_lagrange_basis trusted_setup.rs:24-49; the same table is the eq(point, .) of the sparse GKR prover).
//
// The reference doubles the table once per variable.  Level by level that is one launch per variable, and the
// first dozen launches are latency only (measured: 22 launches x ~7.6 us for a 2^22 table).  Here the table is the
// outer product of the tables of its high and low variable halves: a 2^22 table is ONE small launch that builds both
// 11-variable halves (each itself the outer product, in LDS, of two <= 6-variable tables) and ONE outer-product kernel that
// does all the real work: one multiplication and one 32-byte store per entry.
// Products of the same factors in another order are the same field element: the table is unchanged bit for bit.
#pragma once
#include <vector>

#include "context.h"
#include "mle_kernels.cuh"

namespace zk {

constexpr int kEqDirectBits = 6;

// out[i] = hi[i >> lbits] * lo[i & (2^lbits - 1)]
template <class F> __global__ void eq_outer_kernel(const void *__restrict__ hi, const void *__restrict__ lo, unsigned lbits, size_t n,
                                                   void *__restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, mask = ((size_t)1 << lbits) - 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        fe_store<F>(out, i, fe_mul<F>(fe_load<F>(hi, i >> lbits), fe_load<F>(lo, i & mask)));
}

// Tables of up to 2 kEqDirectBits = 12 variables in ONE launch by one workgroup: the two <= 6-variable half tables in LDS (a chain of
// <= 6 products), then one product per entry.  blockIdx.x selects one of up to two independent tables (the halves of a larger one).
constexpr int kEqSmallBits = 2 * kEqDirectBits;
template <class F> struct EqSmallArgs {
    Fe<F> tau[2][kEqSmallBits];
    int nbits[2];
    void *out[2];
    Fe<F> scale[2];          // table `which` times scale[which] when scaled[which] (folded into its high half table: 64 products)
    int scaled[2];
    const void *tau_dev[2];  // non-null: variable i of table `which` is element i * tau_stride of this device array (challenges an
    size_t tau_stride;       // earlier kernel of the stream wrote into the proof slots), and the scale is *scale_dev[which]
    const void *scale_dev[2];
};
template <class F> __global__ void __launch_bounds__(1024) eq_small_kernel(EqSmallArgs<F> a) {
    __shared__ Fe<F> th[1 << kEqDirectBits], tl[1 << kEqDirectBits];
    const int which = blockIdx.x, nbits = a.nbits[which];
    const int hbits = nbits / 2, lbits = nbits - hbits;        // variables 0..hbits-1 are the high index bits (MSB first)
    const unsigned tid = threadIdx.x;
    if (tid < 128) {
        const bool low = tid >= 64;
        const unsigned idx = tid & 63u;
        const int nb = low ? lbits : hbits, off = low ? hbits : 0;
        if (idx < (1u << nb)) {
            Fe<F> acc = fe_one<F>();
#pragma unroll 1
            for (int i = 0; i < nb; i++) {
                const Fe<F> t = a.tau_dev[which] ? fe_load<F>(a.tau_dev[which], (size_t)(off + i) * a.tau_stride) : a.tau[which][off + i];
                const bool bit = (idx >> (nb - 1 - i)) & 1u;
                acc = fe_mul<F>(acc, bit ? t : fe_sub<F>(fe_one<F>(), t));
            }
            if (!low && a.scaled[which]) acc = fe_mul<F>(acc, a.scale_dev[which] ? fe_load<F>(a.scale_dev[which], 0) : a.scale[which]);
            (low ? tl : th)[idx] = acc;
        }
    }
    __syncthreads();
    const unsigned n = 1u << nbits, mask = (1u << lbits) - 1u;
    for (unsigned e = tid; e < n; e += blockDim.x) fe_store<F>(a.out[which], e, fe_mul<F>(th[e >> lbits], tl[e & mask]));
}

// Enqueue the construction of the 2^nbits-entry table of `point` (nbits elements, u64 limbs) into `out` (device).
// Temporaries come from the caching pool and are returned to it when the caller calls release() -- after the
// stream has been synchronised or further work on the same stream has been enqueued (the pool is stream-ordered).
// A table of <= 12 variables is one launch; one of <= 24 variables is two: both halves by eq_small_kernel, then the outer
// product (r1: the recursion down to 6-variable direct kernels took 7 dependent launches per 22-variable table, ~45 us of
// launch latency before the one kernel that does the work).
template <class F> struct EqBuilder {
    std::vector<void *> temps;
    ~EqBuilder() { release(); }
    void release() {
        for (void *p : temps) pool_free(p);
        temps.clear();
    }
    static void load_taus(Fe<F> *dst, const uint64_t *point, uint32_t nbits) {
        for (int i = 0; i < kEqSmallBits; i++) {
            if ((uint32_t)i < nbits) memcpy(dst[i].l, point + (size_t)i * (F::N / 2), 4 * F::N);
            else dst[i] = fe_zero<F>();
        }
    }
    // `scale` (may be null): the table times that constant, i.e. out[i] = scale * eq(point, i) -- the constant rides on the high
    // half table, so a weighted sum alpha eq(rb, .) + beta eq(rc, .) costs its consumers no products (zkmle_gkr_sparse.hip)
    // the same from a point that lives in device memory (element i at dev + i * stride elements; scale_dev: one element or null)
    // only the two half tables of a table of > kEqSmallBits variables (entry i of the full table = hi[i >> lbits] * lo[i & (2^lbits - 1)]):
    // for consumers that gather a few entries each and can afford the product (zkmle_gkr_sparse.hip), 2 x 64 KB that stay in L2 instead of
    // a 2^nbits-entry table written once and gathered from HBM.  The pointers stay valid until release().
    int halves_dev(const void *dev, size_t stride, uint32_t nbits, const void **hi_out, const void **lo_out, unsigned *lbits_out,
                   const void *scale_dev = nullptr) {
        const size_t esz = 4 * F::N;
        const uint32_t hbits = nbits / 2, lbits = nbits - hbits;
        void *hi = nullptr, *lo = nullptr;
        ZK_TRY(pool_alloc(((size_t)1 << hbits) * esz, &hi));
        temps.push_back(hi);
        ZK_TRY(pool_alloc(((size_t)1 << lbits) * esz, &lo));
        temps.push_back(lo);
        const void *dev_lo = (const char *)dev + (size_t)hbits * stride * esz;
        if (lbits <= (uint32_t)kEqSmallBits) {
            EqSmallArgs<F> a{};
            a.nbits[0] = (int)hbits; a.nbits[1] = (int)lbits;
            a.out[0] = hi; a.out[1] = lo;
            a.tau_dev[0] = dev; a.tau_dev[1] = dev_lo; a.tau_stride = stride;
            a.scaled[0] = scale_dev ? 1 : 0; a.scale_dev[0] = scale_dev;
            eq_small_kernel<F><<<2, 1024, 0, cur_stream()>>>(a);
            ZK_HIP(hipGetLastError());
        } else {
            ZK_TRY(build_dev(dev, stride, hbits, hi, scale_dev));
            ZK_TRY(build_dev(dev_lo, stride, lbits, lo));
        }
        *hi_out = hi; *lo_out = lo; *lbits_out = lbits;
        return ZK_OK;
    }
    int build_dev(const void *dev, size_t stride, uint32_t nbits, void *out, const void *scale_dev = nullptr) {
        if (nbits <= (uint32_t)kEqSmallBits) {
            EqSmallArgs<F> a{};
            a.nbits[0] = (int)nbits; a.nbits[1] = 0;
            a.out[0] = out; a.out[1] = nullptr;
            a.tau_dev[0] = dev; a.tau_dev[1] = nullptr; a.tau_stride = stride;
            a.scaled[0] = scale_dev ? 1 : 0; a.scale_dev[0] = scale_dev;
            eq_small_kernel<F><<<1, 1024, 0, cur_stream()>>>(a);
            ZK_HIP(hipGetLastError());
            return ZK_OK;
        }
        const void *hi = nullptr, *lo = nullptr;
        unsigned lbits = 0;
        ZK_TRY(halves_dev(dev, stride, nbits, &hi, &lo, &lbits, scale_dev));
        const size_t n = (size_t)1 << nbits;
        eq_outer_kernel<F><<<grid_for(n), kBlock, 0, cur_stream()>>>(hi, lo, lbits, n, out);
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }
    int halves(const uint64_t *point, uint32_t nbits, const void **hi_out, const void **lo_out, unsigned *lbits_out, const Fe<F> *scale = nullptr) {
        const uint32_t hbits = nbits / 2, lbits = nbits - hbits;       // variables 0..hbits-1 are the high index bits
        void *hi = nullptr, *lo = nullptr;
        ZK_TRY(pool_alloc(((size_t)1 << hbits) * 4 * F::N, &hi));
        temps.push_back(hi);
        ZK_TRY(pool_alloc(((size_t)1 << lbits) * 4 * F::N, &lo));
        temps.push_back(lo);
        if (lbits <= (uint32_t)kEqSmallBits) {                         // both halves in one launch
            EqSmallArgs<F> a{};
            load_taus(a.tau[0], point, hbits);
            load_taus(a.tau[1], point + (size_t)hbits * (F::N / 2), lbits);
            a.nbits[0] = (int)hbits; a.nbits[1] = (int)lbits;
            a.out[0] = hi; a.out[1] = lo;
            a.scaled[0] = scale ? 1 : 0; a.scaled[1] = 0;
            a.scale[0] = scale ? *scale : fe_zero<F>(); a.scale[1] = fe_zero<F>();
            eq_small_kernel<F><<<2, 1024, 0, cur_stream()>>>(a);
            ZK_HIP(hipGetLastError());
        } else {
            ZK_TRY(build(point, hbits, hi, scale));
            ZK_TRY(build(point + (size_t)hbits * (F::N / 2), lbits, lo));
        }
        *hi_out = hi; *lo_out = lo; *lbits_out = lbits;
        return ZK_OK;
    }
    int build(const uint64_t *point, uint32_t nbits, void *out, const Fe<F> *scale = nullptr) {
        if (nbits <= (uint32_t)kEqSmallBits) {
            EqSmallArgs<F> a{};
            load_taus(a.tau[0], point, nbits);
            load_taus(a.tau[1], point, 0);
            a.nbits[0] = (int)nbits; a.nbits[1] = 0;
            a.out[0] = out; a.out[1] = nullptr;
            a.scaled[0] = scale ? 1 : 0; a.scaled[1] = 0;
            a.scale[0] = scale ? *scale : fe_zero<F>(); a.scale[1] = fe_zero<F>();
            eq_small_kernel<F><<<1, 1024, 0, cur_stream()>>>(a);
            ZK_HIP(hipGetLastError());
            return ZK_OK;
        }
        const void *hi = nullptr, *lo = nullptr;
        unsigned lbits = 0;
        ZK_TRY(halves(point, nbits, &hi, &lo, &lbits, scale));
        const size_t n = (size_t)1 << nbits;
        eq_outer_kernel<F><<<grid_for(n), kBlock, 0, cur_stream()>>>(hi, lo, lbits, n, out);
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }
};

}  // namespace zk
