// msm_bucket.hip -- the hot Pippenger kernel: per-bucket-segment XYZZ accumulation (own TU: compile time).
#include "context.h"
#include "msm_kernels.cuh"
#include "g1u.cuh"

namespace zk {

// the bucket whose range [starts[b], starts[b + 1]) holds position `pos` (< starts[nbuckets]): the last b with starts[b] <= pos
__device__ __forceinline__ size_t msm_bucket_of(const uint64_t *__restrict__ starts, size_t nbuckets, uint64_t pos) {
    size_t lo = 0, hi = nbuckets;
    while (hi - lo > 1) {
        const size_t mid = (lo + hi) >> 1;
        if (starts[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

// step 3: one lane per RUN of `run` consecutive entries of the bucket-sorted array, whatever buckets the run crosses: every lane of
// a wave does exactly `run` mixed additions (r3: one lane per bucket segment left the lanes of a wave with different counts -- a
// bucket of 512 +- 22 entries is 8 full segments and a stub, buckets of ~100 entries are 2 segments and a stub -- 6 % of the lane
// slots idle at c = 16 and 20-50 % with the sparser buckets of wider windows).  At a bucket boundary the lane stores what it has as
// one partial sum of the bucket it leaves and starts over.  Partial slots: bucket b owns seg_starts[b] .. seg_starts[b + 1), one
// per run that overlaps it, in run order (msm_kernels.cuh SegFromRuns); the lane that holds the k-th overlapping run writes slot k.
// The accumulator lives in VGPRs in the unsaturated 29-bit form (g1u.cuh); bases are read pre-converted, one 128-byte line per
// point, and the next entry's point is fetched while the current addition runs.
__global__ void __launch_bounds__(256) msm_bucket_sum_kernel(const void *__restrict__ bases_u, const uint32_t *__restrict__ sorted,
                                                             const uint64_t *__restrict__ starts,
                                                             const uint32_t *__restrict__ seg_starts, size_t nbuckets,
                                                             unsigned run, uint64_t entries, void *__restrict__ partials) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t first = t * run;
    if (first >= entries) return;
    const uint64_t end = first + run < entries ? first + run : entries;
    size_t b = msm_bucket_of(starts, nbuckets, first);
    uint64_t next = starts[b + 1];                              // where bucket b ends
    uint32_t slot = seg_starts[b] + (uint32_t)(t - starts[b] / run);
    // what the NEXT boundary will need is loaded one boundary ahead: a lane that crosses a boundary holds up its whole wave, and three
    // dependent loads there (r3, first form of this kernel) cost 10-40 % of the kernel with buckets of 100 / 20 entries
    size_t bn = b + 2 <= nbuckets ? b + 2 : nbuckets;
    uint64_t next2 = starts[bn];                                // where bucket b + 1 ends
    uint32_t slot2 = seg_starts[bn - 1];                        // first slot of bucket b + 1
    G1XyzzU acc;
    acc.inf = true;
    acc.x = u_zero<Fq381>(); acc.y = u_zero<Fq381>(); acc.zz = u_zero<Fq381>(); acc.zzz = u_zero<Fq381>();
    uint32_t v = sorted[first];
    G1AffineU p = g1u_load_affine(bases_u, v & 0x7fffffffu);
    for (uint64_t e = first; e < end; e++) {
        uint32_t vn = 0;
        G1AffineU pn = p;
        if (e + 1 < end) {                                      // prefetch the next point
            vn = sorted[e + 1];
            pn = g1u_load_affine(bases_u, vn & 0x7fffffffu);
        }
        if (e == next) {                                        // bucket boundary inside the run
            g1u_store_xyzz(partials, slot, acc);
            acc.inf = true;
            b++;
            next = next2;
            slot = slot2;                                       // this run is the first to overlap the new bucket
            if (next <= e) {                                    // empty buckets in between: find the one that holds e
                b = msm_bucket_of(starts, nbuckets, e);
                next = starts[b + 1];
                slot = seg_starts[b];
            }
            bn = b + 2 <= nbuckets ? b + 2 : nbuckets;
            next2 = starts[bn];
            slot2 = seg_starts[bn - 1];
        }
        g1u_madd(acc, p, (v >> 31) != 0);
        v = vn;
        p = pn;
    }
    g1u_store_xyzz(partials, slot, acc);                        // internal form: the combination and reduction kernels stay in it
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
                          size_t nbuckets, unsigned run, uint64_t entries, void *partials, hipStream_t s) {
    const uint64_t lanes = (entries + run - 1) / run;
    if (lanes == 0) return ZK_OK;
    msm_bucket_sum_kernel<<<(unsigned)((lanes + 255) / 256), 256, 0, s>>>(bases, sorted, starts, seg_starts, nbuckets, run, entries, partials);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int launch_g1_bases_to_u(const void *affine, size_t n, void *out_u, hipStream_t s) {
    g1_bases_to_u_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(affine, n, out_u);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

}  // namespace zk
