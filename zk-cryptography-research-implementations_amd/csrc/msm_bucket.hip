// msm_bucket.hip -- the hot Pippenger kernel: per-bucket-segment XYZZ accumulation (own TU: compile time).
#include "context.h"
#include "msm_kernels.cuh"

namespace zk {

// step 3: one lane per segment of at most seg_len entries of one bucket
__global__ void __launch_bounds__(256) msm_bucket_sum_kernel(const void *__restrict__ bases, const uint32_t *__restrict__ sorted,
                                                             const uint64_t *__restrict__ starts,
                                                             const uint32_t *__restrict__ seg_starts, size_t nbuckets,
                                                             unsigned seg_len, uint32_t nseg, void *__restrict__ partials) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nseg) return;
    // bucket of segment t: last index with seg_starts[idx] <= t (binary search; empty buckets own no segment)
    size_t lo = 0, hi = nbuckets;
    while (hi - lo > 1) {
        size_t mid = (lo + hi) >> 1;
        if (seg_starts[mid] <= t) lo = mid; else hi = mid;
    }
    size_t b = lo;
    uint64_t first = starts[b] + (uint64_t)(t - seg_starts[b]) * seg_len;
    uint64_t end = starts[b + 1];
    if (first + seg_len < end) end = first + seg_len;
    G1Xyzz acc = g1_xyzz_inf();
    for (uint64_t e = first; e < end; e++) {
        uint32_t v = sorted[e];
        G1Affine p = g1_load_affine(bases, v & 0x7fffffffu);
        if (v >> 31) p.y = fe_neg<Fq>(p.y);
        acc = g1_madd(acc, p);
    }
    g1_store_xyzz(partials, t, acc);
}


int launch_msm_bucket_sum(const void *bases, const uint32_t *sorted, const uint64_t *starts, const uint32_t *seg_starts,
                          size_t nbuckets, unsigned seg_len, uint32_t nseg, void *partials, hipStream_t s) {
    msm_bucket_sum_kernel<<<(nseg + 255) / 256, 256, 0, s>>>(bases, sorted, starts, seg_starts, nbuckets, seg_len, nseg, partials);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

}  // namespace zk
