// msm_bucket.hip -- the hot Pippenger kernel: per-bucket-segment XYZZ accumulation (own TU: compile time).
#include "context.h"
#include "msm_kernels.cuh"
#include "g1u.cuh"

namespace zk {

// step 3: one lane per segment of at most seg_len entries of one bucket.  The accumulator lives in VGPRs in
// the unsaturated 29-bit form (g1u.cuh); bases are read pre-converted, one 128-byte line per point, and the
// next entry's point is fetched while the current addition runs.
__global__ void __launch_bounds__(256) msm_bucket_sum_kernel(const void *__restrict__ bases_u, const uint32_t *__restrict__ sorted,
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
    G1XyzzU acc;
    acc.inf = true;
    acc.x = u_zero<Fq381>(); acc.y = u_zero<Fq381>(); acc.zz = u_zero<Fq381>(); acc.zzz = u_zero<Fq381>();
    if (first < end) {
        uint32_t v = sorted[first];
        G1AffineU p = g1u_load_affine(bases_u, v & 0x7fffffffu);
        for (uint64_t e = first; e < end; e++) {
            uint32_t vn = 0;
            G1AffineU pn = p;
            if (e + 1 < end) {                                  // prefetch the next point
                vn = sorted[e + 1];
                pn = g1u_load_affine(bases_u, vn & 0x7fffffffu);
            }
            g1u_madd(acc, p, (v >> 31) != 0);
            v = vn;
            p = pn;
        }
    }
    g1u_store_xyzz(partials, t, acc);                       // internal form: the combination and reduction kernels stay in it
}

// stored affine bases (96 B) -> internal form (128 B per point)
__global__ void g1_bases_to_u_kernel(const void *__restrict__ affine, size_t n, void *__restrict__ out_u) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine p = g1_load_affine(affine, i);
    uint32_t *o = reinterpret_cast<uint32_t *>(out_u) + i * (2 * kUWords);
    fqu_store(o, u_from_std<Fq381>(p.x));                       // 0 stays exactly 0: infinity keeps its encoding
    fqu_store(o + kUWords, u_from_std<Fq381>(p.y));
}

int launch_msm_bucket_sum(const void *bases, const uint32_t *sorted, const uint64_t *starts, const uint32_t *seg_starts,
                          size_t nbuckets, unsigned seg_len, uint32_t nseg, void *partials, hipStream_t s) {
    msm_bucket_sum_kernel<<<(nseg + 255) / 256, 256, 0, s>>>(bases, sorted, starts, seg_starts, nbuckets, seg_len, nseg, partials);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int launch_g1_bases_to_u(const void *affine, size_t n, void *out_u, hipStream_t s) {
    g1_bases_to_u_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(affine, n, out_u);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

}  // namespace zk
