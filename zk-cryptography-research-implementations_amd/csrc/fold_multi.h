// fold_multi.h -- host-side launcher of foldk_seg_sums_kernel (mle_kernels.cuh): several variables of a table folded in one pass.
#pragma once
#include <stdlib.h>
#include "context.h"
#include "mle_kernels.cuh"

namespace zk {

inline unsigned multi_blocks() {                       // ZK_MULTI_BLOCKS overrides kMultiBlocks, for measurements
    static const unsigned v = [] { const char *e = getenv("ZK_MULTI_BLOCKS"); int k = e ? atoi(e) : kMultiBlocks; return (unsigned)(k < 256 ? 256 : (k > 16384 ? 16384 : k)); }();
    return v;
}
inline unsigned multi_bps(size_t seglen, int m) {
    size_t b = (seglen + kBlock - 1) / kBlock;
    const size_t cap = (size_t)multi_blocks() >> m;
    return (unsigned)(b > cap ? cap : b < 1 ? 1 : b);
}
// fold the k variables whose values sit at rp[0 .. k) (device memory) and leave 2^m_next segment sums of the output (m_next = 0: none).
// FIN (translation units that include basic_multi.cuh): with `fin`, the pass's last workgroup runs the exchange on those sums itself.
template <class F, bool FIN = false>
int launch_foldk(const void *in, void *out, size_t n, int k, const void *const *rp, int m_next, void *part, unsigned *bps_out, const MultiFin *fin = nullptr) {
    FoldKArgs a{};
    a.in = in; a.out = out; a.n = n;
    for (int i = 0; i < k; i++) a.r[i] = rp[i];
    a.partials = m_next ? part : nullptr;
    const MultiFin f = (FIN && fin && m_next) ? *fin : MultiFin{};
    if constexpr (FIN) {
        // an output of fewer than 4 waves per SIMD (1024 workgroups) with one lane per output: two lanes per output (foldk_seg_sums_split2_kernel)
        static const bool split2_on = [] { const char *e = getenv("ZK_FOLD_SPLIT2"); return !(e && e[0] == '0'); }();
        const size_t seglen = n >> m_next, half_block = kBlock / 2;
        if (split2_on && k >= 5 && m_next && seglen >= half_block && n / kBlock < (size_t)multi_blocks()) {
            size_t b = seglen / half_block;                      // powers of two: seglen is a multiple of bps * kBlock / 2
            const size_t cap = (size_t)multi_blocks() >> m_next;
            while (b > cap && b > 1) b >>= 1;                    // stays a power of two whatever ZK_MULTI_BLOCKS says
            a.bps = (unsigned)b;
            const unsigned grid2 = a.bps << m_next;
            if (k == 5) foldk_seg_sums_split2_kernel<F, 5, true><<<grid2, kBlock, 0, cur_stream()>>>(a, f);
            else if (k == 6) foldk_seg_sums_split2_kernel<F, 6, true><<<grid2, kBlock, 0, cur_stream()>>>(a, f);
            else if (k == 7) foldk_seg_sums_split2_kernel<F, 7, true><<<grid2, kBlock, 0, cur_stream()>>>(a, f);
            else if (k == 8) foldk_seg_sums_split2_kernel<F, 8, true><<<grid2, kBlock, 0, cur_stream()>>>(a, f);
            else return ZK_E_ARG;
            ZK_HIP(hipGetLastError());
            if (bps_out) *bps_out = a.bps;
            return ZK_OK;
        }
    }
    a.bps = multi_bps(n >> m_next, m_next);
    const unsigned grid = a.bps << m_next;
    switch (k) {
        case 1: foldk_seg_sums_kernel<F, 1, FIN><<<grid, kBlock, 0, cur_stream()>>>(a, f); break;
        case 2: foldk_seg_sums_kernel<F, 2, FIN><<<grid, kBlock, 0, cur_stream()>>>(a, f); break;
        case 3: foldk_seg_sums_kernel<F, 3, FIN><<<grid, kBlock, 0, cur_stream()>>>(a, f); break;
        case 4: foldk_seg_sums_kernel<F, 4, FIN><<<grid, kBlock, 0, cur_stream()>>>(a, f); break;
        case 5: case 6: case 7: case 8:
            if constexpr (FIN) {                            // more than four variables per pass: the basic sumcheck only (zkmle_sumcheck.hip)
                if (k == 5) foldk_seg_sums_kernel<F, 5, true><<<grid, kBlock, 0, cur_stream()>>>(a, f);
                else if (k == 6) foldk_seg_sums_kernel<F, 6, true><<<grid, kBlock, 0, cur_stream()>>>(a, f);
                else if (k == 7) foldk_seg_sums_kernel<F, 7, true><<<grid, kBlock, 0, cur_stream()>>>(a, f);
                else foldk_seg_sums_kernel<F, 8, true><<<grid, kBlock, 0, cur_stream()>>>(a, f);
                break;
            }
            return ZK_E_ARG;
        default: return ZK_E_ARG;
    }
    ZK_HIP(hipGetLastError());
    if (bps_out) *bps_out = a.bps;
    return ZK_OK;
}

// a pass with a short output and no segment sums: kFoldSplit lanes per output (foldk_split_kernel)
template <class F> int launch_foldk_split(const void *in, void *out, size_t n, int k, const void *const *rp) {
    FoldKArgs a{};
    a.in = in; a.out = out; a.n = n;
    for (int i = 0; i < k; i++) a.r[i] = rp[i];
    const unsigned grid = (unsigned)((n + kFoldSplitBlock / kFoldSplit - 1) / (kFoldSplitBlock / kFoldSplit));
    switch (k) {
        case 4: foldk_split_kernel<F, 4><<<grid, kFoldSplitBlock, 0, cur_stream()>>>(a); break;
        case 5: foldk_split_kernel<F, 5><<<grid, kFoldSplitBlock, 0, cur_stream()>>>(a); break;
        case 6: foldk_split_kernel<F, 6><<<grid, kFoldSplitBlock, 0, cur_stream()>>>(a); break;
        case 7: foldk_split_kernel<F, 7><<<grid, kFoldSplitBlock, 0, cur_stream()>>>(a); break;
        case 8: foldk_split_kernel<F, 8><<<grid, kFoldSplitBlock, 0, cur_stream()>>>(a); break;
        default: return ZK_E_ARG;
    }
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

}  // namespace zk
