// zkmle_kzg.hip -- C ABI: G1 bases in HBM, Pippenger MSM, and the multilinear-KZG prover side
// (trusted-setup G1 powers, commit, open).  Mirrors multilinear_kzg/src/{multilinear_kzg,trusted_setup}.rs;
// pairing-based `verify` (G2 / GT) is out of scope.
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include "context.h"
#define ZK_MSM_LIGHT_KERNELS
#include "eq_table.cuh"
#include "msm_kernels.cuh"
#include "msm_sort_wide.cuh"

using namespace zk;

struct zk_g1_bases {
    size_t n;
    void *dptr;            // n affine points, 96 B each (stored Montgomery form: what upload / download see)
    void *dptr_u = nullptr; // the same points pre-converted for the bucket kernel (128 B each), built on first use
    // zk_g1_bases_precompute: one pre-converted copy of the points per window, [w][i] = 2^(pre_c w) B_i (w < pre_nwin), so that every
    // window of an MSM feeds ONE bucket set (msm_core, `pre`)
    void *pre_u = nullptr;
    int pre_c = 0;
    unsigned pre_nwin = 0;
    // prove_succinct opens twice on the same powers at every call: the pre-summed levels (zk_kzg_opening_key_new, milliseconds even for 2^8 points) are
    // built at the first one and kept with the points they are sums of
    struct zk_kzg_opening_key *own_key = nullptr;
};
struct zk_kzg_opening_key {
    // level[t] (t = 1..nvars): 2^(nvars - t) pre-summed affine bases B^(t)_k = sum_{h < 2^t} B_{h 2^(nvars-t) + k}
    std::vector<zk_g1_bases *> level;
    size_t nvars;
    // the levels of <= 2^open_batch_bits() bases once more, pre-converted, one copy per window: level j = t - small_t0,
    // so that their MSMs run as ONE batched pass (msm_core)
    void *small_u = nullptr;      // [window][off_j + k] = 2^(small_c * window) B^(small_t0 + j)_k, pre-converted (msm_core, shifted);
                                  // the levels lie end to end, off_j = 2^(small_bits + 1) - 2^(small_bits + 1 - j)
    unsigned small_bits = 0;      // the first batched level has 2^small_bits points
    size_t small_t0 = 0;          // first batched level
    int small_c = 0;              // window size of the batched pass
};
// levels of at most 2^open_batch_bits() bases are batched.  Default 19: 2^20 points x 16 windows x 128 B = 2 GiB per opening key
// at most, +30 ms to build it; measured r1 (open_and_prove, ms, bits = 13 / 16 / 19): 2^16 7.0 / 3.3 / 3.4, 2^20 21.6 / 15.9 / 10.5,
// 2^24 87 / 81 / 77 (profiles/r1/bench_kzg_e2e.jsonl).  ZK_KZG_OPEN_BATCH_BITS overrides, for measurements.
static unsigned open_batch_bits() {
    static const unsigned v = [] {
        const char *e = getenv("ZK_KZG_OPEN_BATCH_BITS");
        int k = e ? atoi(e) : 19;
        return (unsigned)(k < 1 ? 1 : (k > 22 ? 22 : k));
    }();
    return v;
}

namespace {

struct DevBuf {     // per-call scratch from the caching pool (context.h)
    void *p = nullptr;
    ~DevBuf() { pool_free(p); }
    int alloc(size_t bytes) { return pool_alloc(bytes, &p); }
    void release() { pool_free(p); p = nullptr; }
};
struct Events {
    hipEvent_t e[8];
    int n = 0;
    ~Events() { for (int i = 0; i < n; i++) (void)hipEventDestroy(e[i]); }
    int mark() { ZK_HIP(hipEventCreate(&e[n])); ZK_HIP(hipEventRecord(e[n], cur_stream())); n++; return ZK_OK; }
    float ms(int a, int b) { float v = 0; (void)hipEventElapsedTime(&v, e[a], e[b]); return v; }
};

template <class F> Fe<F> load_el(const uint64_t *src) { Fe<F> e; memcpy(e.l, src, 4 * F::N); return e; }

void affine_to_u64(const G1Affine &p, uint64_t *out12) {
    memcpy(out12, p.x.l, 48);
    memcpy(out12 + 6, p.y.l, 48);
}
G1Affine affine_from_u64(const uint64_t *in12) {
    G1Affine p;
    memcpy(p.x.l, in12, 48);
    memcpy(p.y.l, in12 + 6, 48);
    return p;
}

// window size: 255-bit scalars split evenly for c = 16 (top window 15 bits = a full signed digit);
// smaller problems take smaller windows so the bucket reduction does not dominate
int pick_window(size_t n) {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    if (lg >= 19) return 16;       // measured r1 (ms, c = 13 / 16): 2^18 2.96 / 3.53, 2^20 6.35 / 6.05, 2^22 24.7 / 16.6
    if (lg >= 14) return 13;
    int c = lg - 3;
    return c < 4 ? 4 : c;
}

int bases_u(const zk_g1_bases *b, const void **out);

// `batch` independent MSMs in ONE pass of the pipeline.  Small MSMs are pure latency (a 2^12-term MSM takes ~1.6 ms of dependent launches and single-lane addition
// chains whatever its size, plus a serial window combination), so the n MSMs of a KZG opening share one pass.
// shifted = false (batch must be 1): the plain MSM of n_sub terms, d_bases = the pre-converted points.
// shifted = true: the MSMs are the halving levels of a KZG opening laid end to end, MSM j = 2^(batch-1-j) terms starting at
// 2^batch - 2^(batch-j) (n_sub = the first level's size = 2^(batch-1)), and d_bases holds one copy of the points per
// window, [w][i] = 2^(c w) B_i (c must be given): every window of MSM j feeds the SAME bucket set, so there are `batch` bucket
// sets, one reduction each and no window combination.
// results: `batch` XYZZ points on the host.
// wide-window counting sort (msm_sort_wide.cuh): digits -> sorted entries (in `e1`) + per-bucket counts
int msm_sort_wide(const void *d_scalars, size_t n, const WidePlan &pl, DevBuf &digits, DevBuf &e1, uint32_t *d_totals) {
    const size_t cap = (size_t)pl.nwin * n;                 // entries, zero digits included
    DevBuf k1, k2, hist1, off1, pstart1, hist2, off2, pstart2, slice_tot;
    ZK_TRY(digits.alloc(cap * 4));                          // u32 digits; reused as the level-2 entry array
    ZK_TRY(e1.alloc(cap * 4));                              // level-1 entries; reused as the final order
    ZK_TRY(k1.alloc(cap * 2));
    const size_t n1 = (size_t)pl.ctot1 * pl.nbins1;
    ZK_TRY(hist1.alloc(n1 * 4));
    ZK_TRY(off1.alloc(n1 * 8));
    ZK_TRY(pstart1.alloc(((size_t)pl.nbins1 + 1) * 8));
    msmw_digits_hist_kernel<<<pl.nchunks1, kWideBlock, ((size_t)pl.nwin << pl.hb1) * 4, cur_stream()>>>(d_scalars, n, pl, (uint32_t *)digits.p, (uint32_t *)hist1.p);
    {
        const unsigned nslices = (pl.ctot1 + kWideScanRows - 1) / kWideScanRows;
        const dim3 grid((pl.nbins1 + 255) / 256, nslices);
        ZK_TRY(slice_tot.alloc((size_t)nslices * pl.nbins1 * 8));
        msmw_scan1_sums_kernel<<<grid, 256, 0, cur_stream()>>>((const uint32_t *)hist1.p, pl.nbins1, pl.ctot1, (uint64_t *)slice_tot.p);
        msmw_scan1_bins_kernel<<<1, kWideBlock, 0, cur_stream()>>>((uint64_t *)slice_tot.p, pl.nbins1, nslices, (uint64_t *)pstart1.p);
        msmw_scan1_offsets_kernel<<<grid, 256, 0, cur_stream()>>>((const uint32_t *)hist1.p, pl.nbins1, pl.ctot1, (const uint64_t *)slice_tot.p,
                                                                  (const uint64_t *)pstart1.p, (uint64_t *)off1.p);
    }
    msmw_l1_scatter_kernel<<<pl.nwin * pl.nchunks1, kWideBlock, 0, cur_stream()>>>((const uint32_t *)digits.p, n, pl, (const uint64_t *)off1.p, (uint32_t *)e1.p,
                                                                                 (uint16_t *)k1.p);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipFuncSetAttribute((const void *)msmw_l3_kernel<uint8_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWideFineLds));
    ZK_HIP(hipFuncSetAttribute((const void *)msmw_l3_kernel<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWideFineLds));
    if (pl.mb > 0) {
        const size_t n2 = (size_t)pl.ngroups * pl.c2;
        ZK_TRY(k2.alloc(cap));
        ZK_TRY(hist2.alloc(n2 * 4));
        ZK_TRY(off2.alloc(n2 * 8));
        ZK_TRY(pstart2.alloc((pl.ngroups + 1) * 8));
        msmw_l2_hist_kernel<<<pl.nbins1 * pl.c2, 256, 0, cur_stream()>>>((const uint16_t *)k1.p, (const uint64_t *)pstart1.p, pl, (uint32_t *)hist2.p);
        msmw_scan2_kernel<<<pl.nbins1, 256, 0, cur_stream()>>>((const uint32_t *)hist2.p, (const uint64_t *)pstart1.p, pl, (uint64_t *)off2.p, (uint64_t *)pstart2.p);
        msmw_l2_scatter_kernel<<<pl.nbins1 * pl.c2, kWideBlock, 0, cur_stream()>>>((const uint32_t *)e1.p, (const uint16_t *)k1.p, (const uint64_t *)pstart1.p, pl,
                                                                                  (const uint64_t *)off2.p, (uint32_t *)digits.p, (uint8_t *)k2.p);
        msmw_l3_kernel<uint8_t><<<(unsigned)pl.ngroups, kWideBlock, kWideFineLds, cur_stream()>>>((const uint32_t *)digits.p, (const uint8_t *)k2.p, (const uint64_t *)pstart2.p,
                                                                                                 pl.lb3, (uint32_t *)e1.p, d_totals);
    } else {
        // no middle level: a level-1 part is a group; level 3 must not write where it still reads, so the final order goes to `digits`
        msmw_l3_kernel<uint16_t><<<(unsigned)pl.ngroups, kWideBlock, kWideFineLds, cur_stream()>>>((const uint32_t *)e1.p, (const uint16_t *)k1.p, (const uint64_t *)pstart1.p,
                                                                                                  pl.lb3, (uint32_t *)digits.p, d_totals);
        void *t = digits.p; digits.p = e1.p; e1.p = t;      // the caller finds the final order in `e1`
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(cur_stream()));             // the intermediates are freed on scope exit
    return ZK_OK;
}

int msm_core(const void *d_scalars, const void *d_bases, size_t n_sub, unsigned batch, bool shifted, int c, G1Xyzz *result,
             zk_msm_stats *stats, bool pre = false) {
    if (batch == 0 || (!shifted && batch != 1) || (shifted && n_sub != ((size_t)1 << (batch - 1))) || (shifted && pre)) return ZK_E_ARG;
    const size_t n = shifted ? ((size_t)1 << batch) - 1 : n_sub;   // digit stream: [window][i]
    if (c == 0) c = pick_window(n_sub);
    // wide windows (msm_sort_wide.cuh): c > 16 -- more buckets than one workgroup's LDS counts -- and every MSM on precomputed
    // window-shifted bases (`pre`: d_bases = [w][i], all windows feed ONE bucket set).  ZK_MSM_WIDE=1 forces the wide sort for c <= 16.
    static const bool force_wide = [] { const char *e = getenv("ZK_MSM_WIDE"); return e && e[0] == '1'; }();
    bool wide = !shifted && (c > 16 || pre || (force_wide && c >= 9));
    if (c < 2 || c > (wide ? 24 : 16)) return ZK_E_ARG;
    const unsigned nwin1 = (256 + c - 1) / c, nb = 1u << (c - 1);
    // (ZK_MSM_WIDE=1 on a window whose first sort level has more bins than the wide sort takes: the LDS sort, which holds any c <= 16)
    if (wide && c <= 16 && !pre && (nwin1 << 7) > 2048) wide = false;
    // A window far wider than the MSM is long asks for bucket and reduction arrays of tens of GB to add a handful of points (c = 24: 11 x 2^23
    // buckets, two arrays of 192 B each: ~35 GB whatever n is).  On this GPU's 288 GB that works (the tests run it); where it would not fit in
    // half of the memory that is free, the call is refused with the reason instead of dying in an allocation.
    if (c > 16 && (size_t)nb > 64 * (n_sub ? n_sub : 1)) {
        size_t free_b = 0, total_b = 0;
        const size_t need = (size_t)(pre ? 1u : nwin1) * nb * 2 * sizeof(G1Xyzz);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > free_b / 2) {
            set_last_error("zk_msm_g1: window_bits this wide for so few terms needs more bucket memory than half of what is free; use 0 (automatic) or a narrower window");
            return ZK_E_ARG;
        }
    }
    if (n * ((shifted || pre) ? nwin1 : 1) >= ((size_t)1 << 31)) return ZK_E_ARG;   // index + sign are packed in 32 bits
    const unsigned nwin = shifted ? batch : pre ? 1u : nwin1;          // bucket sets
    const size_t nbuckets = (size_t)nwin * nb;
    // one (chunk, window) workgroup per CU-slot: long chunks make each workgroup write long runs per bucket
    // (r1: 128 chunks -> scatter 6.4 ms at 2^24; the LDS cursors hold a whole window either way)
    size_t chunk_len = (n + 15) / 16;
    if (chunk_len < 4096) chunk_len = 4096;
    unsigned nchunks = (unsigned)((n + chunk_len - 1) / chunk_len);
    if (shifted) { chunk_len = 0; nchunks = msm_geo_nchunks(batch); }   // pieces of the levels; level j = bucket set j (msm_chunk_range)
    const unsigned by_chunk = shifted ? batch : 0u;
    // one lane per bucket SEGMENT: cap the serial chain so that ~2^22 lanes exist whatever the window size, at least 16 points
    // per lane.  Measured r1 after the bucket update became the unsaturated dual-product form (2^24 terms, bucket phase, ms;
    // lanes 0.8M / 1.3M / 2.4M / 4.4M / 8.6M / 17M): 58.7 / 49.4 / 43.6 / 41.4 / 43.4 / 44.7; minimum 4 / 8 / 16 / 32 points per lane
    // at 2^20 terms: 4.13 / 3.59 / 3.50 / 3.81.  ZK_MSM_SEG_SHIFT / ZK_MSM_SEG_MIN override, for measurements.
    static const int seg_shift = [] { const char *e = getenv("ZK_MSM_SEG_SHIFT"); int k = e ? atoi(e) : 20; return k < 10 ? 10 : (k > 24 ? 24 : k); }();   // r3, balanced runs (2^24, c = 16 / precomputed c = 22, ms; shift 19 / 20 / 21 / 22 / 23): 43.35 / 42.49 / 42.61 / 43.19 / 44.28 and 36.11 / 35.61 / 36.32 / 36.93 / 37.32
    size_t seg_target = ((size_t)n * nwin1) >> seg_shift;
    static const unsigned seg_min = [] { const char *e = getenv("ZK_MSM_SEG_MIN"); int k = e ? atoi(e) : 32; return (unsigned)(k < 2 ? 2 : k); }();   // r3, balanced runs, 2^20 terms (bucket phase, ms; 16 / 32 / 64): 2.97 / 2.86 / 2.86
    unsigned seg_len = (unsigned)(seg_target < seg_min ? seg_min : seg_target);
    // the batched pass feeds a bucket from every window: the largest level's buckets hold ~16 * 2^19 / 2^15 = 256 entries, and
    // 16-entry segments would put them just over the 16-partials-per-bucket limit of the combine kernel (an extra 1.3 ms regroup)
    if (shifted && seg_len < 2 * seg_min) seg_len = 2 * seg_min;
    Events ev;
    ZK_TRY(ev.mark());
    DevBuf digits, hist, totals, starts, seg_starts, sorted, partials, A, R;
    ZK_TRY(totals.alloc(nbuckets * 4));
    ZK_TRY(starts.alloc((nbuckets + 1) * 8));
    ZK_TRY(seg_starts.alloc((nbuckets + 2) * 4));
    size_t lds_bytes = (size_t)nb * 4;
    if (wide) {
        const WidePlan pl = wide_plan(n, (unsigned)c, nwin1, pre);
        if (pl.nbins1 > 2048 || pl.nbuckets != nbuckets) return ZK_E_ARG;
        ZK_TRY(ev.mark());                                   // (the digit kernel takes the first histogram with it: it counts as sort time)
        ZK_TRY(msm_sort_wide(d_scalars, n, pl, digits, sorted, (uint32_t *)totals.p));
    } else {
    ZK_TRY(digits.alloc((size_t)nwin1 * n * 2));
    ZK_TRY(hist.alloc((size_t)nwin1 * nchunks * nb * 4));
    msm_digits_kernel<<<grid_for(n), kBlock, 0, cur_stream()>>>(d_scalars, n, (unsigned)c, nwin1, (uint16_t *)digits.p);
    ZK_HIP(hipGetLastError());
    ZK_TRY(ev.mark());
    // counting sort, bucket counters staged in LDS
    ZK_HIP(hipFuncSetAttribute((const void *)msm_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    ZK_HIP(hipFuncSetAttribute((const void *)msm_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    msm_hist_kernel<<<nwin1 * nchunks, kSortBlock, lds_bytes, cur_stream()>>>((const uint16_t *)digits.p, n, (unsigned)c, nchunks, chunk_len, (uint32_t *)hist.p,
                                                                 by_chunk);
    // per bucket set: exclusive prefix over the blocks that feed it (the chunks of a window; or, shifted, every (chunk, window) of a level)
    if (shifted)
        msm_set_scan_kernel<<<(unsigned)((nbuckets + kBlock - 1) / kBlock), kBlock, 0, cur_stream()>>>((uint32_t *)hist.p, batch, nwin1, nb, (uint32_t *)totals.p);
    else
        msm_chunk_scan_kernel<<<(unsigned)((nbuckets + kBlock - 1) / kBlock), kBlock, 0, cur_stream()>>>((uint32_t *)hist.p, nwin, nchunks, nb,
                                                                                                     (uint32_t *)totals.p);
    }
    const uint32_t *d_totals = (const uint32_t *)totals.p;
    {
        unsigned ntiles = (unsigned)((nbuckets + kScanTile - 1) / kScanTile);
        DevBuf te, ts, tm;
        ZK_TRY(te.alloc((size_t)ntiles * 8));
        ZK_TRY(ts.alloc((size_t)ntiles * 4));
        ZK_TRY(tm.alloc((size_t)ntiles * 4));
        msm_scan_tiles_kernel<<<ntiles, kScanTile, 0, cur_stream()>>>(d_totals, nbuckets, seg_len, (uint64_t *)te.p, (uint32_t *)ts.p, (uint32_t *)tm.p);
        msm_scan_tile_totals_kernel<<<1, 64, 0, cur_stream()>>>((uint64_t *)te.p, (uint32_t *)ts.p, (const uint32_t *)tm.p, ntiles, nbuckets,
                                              (uint64_t *)starts.p, (uint32_t *)seg_starts.p);
        msm_scan_apply_kernel<<<ntiles, kScanTile, 0, cur_stream()>>>(d_totals, nbuckets, seg_len, (const uint64_t *)te.p, (const uint32_t *)ts.p,
                                                     (uint64_t *)starts.p, (uint32_t *)seg_starts.p);
        // partial-sum slots: one per run of seg_len sorted entries that overlaps the bucket (msm_bucket.hip); needs the bucket starts
        ZK_TRY(msm_seg_scan(SegFromRuns{(const uint64_t *)starts.p, seg_len}, nbuckets, (uint32_t *)ts.p, (uint32_t *)tm.p, (uint32_t *)seg_starts.p, cur_stream()));
        ZK_HIP(hipGetLastError());
    }
    // the three totals the host needs to size what follows: both copies into pinned staging, ONE synchronisation (three before: ~60 us of
    // a 2^20-term MSM)
    uint64_t entries = 0;
    uint32_t tail[2] = {0, 0};                              // {segments, largest per-bucket segment count}
    {
        void *stage = nullptr;
        ZK_TRY(host_staging(16, &stage));
        ZK_HIP(hipMemcpyAsync(stage, (uint64_t *)starts.p + nbuckets, 8, hipMemcpyDeviceToHost, cur_stream()));
        ZK_HIP(hipMemcpyAsync((char *)stage + 8, (uint32_t *)seg_starts.p + nbuckets, 8, hipMemcpyDeviceToHost, cur_stream()));
        ZK_HIP(hipStreamSynchronize(cur_stream()));         // (also: the scan's temporaries are freed on scope exit above)
        memcpy(&entries, stage, 8);
        memcpy(tail, (char *)stage + 8, 8);
    }
    uint32_t nseg = tail[0], max_segs = tail[1];
    if (!wide) {
    ZK_TRY(sorted.alloc((entries ? entries : 1) * 4));
    static const int two_level_bits = [] { const char *e = getenv("ZK_MSM_TWO_LEVEL_BITS"); int k = e ? atoi(e) : 20; return k < 12 ? 12 : k; }();
    // measured r1 (sort phase, ms, one-level / two-level): 2^18 0.17 / 0.39, 2^19 0.35 / 0.37, 2^20 0.57 / 0.44, 2^21 0.91 / 0.63,
    // 2^24 8.6 / 3.5
    if (!shifted && c >= 12 && n >= ((size_t)1 << two_level_bits)) {
        // two-level scatter: partition by the high bits of the bucket id (tile-sorted whole-run writes), then finish
        // each 256-bucket partition with one workgroup (msm_kernels.cuh)
        const unsigned lb = (unsigned)(c - 1) < kFineBits ? (unsigned)(c - 1) : kFineBits, nh = 1u << (c - 1 - lb);
        size_t pchunk_len = (n + 255) / 256;
        if (pchunk_len < (size_t)kPartTile) pchunk_len = kPartTile;
        const unsigned pchunks = (unsigned)((n + pchunk_len - 1) / pchunk_len);
        DevBuf phist, poff, part_e, part_l;
        ZK_TRY(phist.alloc((size_t)nwin * nh * pchunks * 4));
        ZK_TRY(poff.alloc((size_t)nwin * nh * pchunks * 8));
        ZK_TRY(part_e.alloc((entries ? entries : 1) * 4));
        ZK_TRY(part_l.alloc(entries ? entries : 1));
        msm_part_hist_kernel<<<nwin * pchunks, 256, 0, cur_stream()>>>((const uint16_t *)digits.p, n, (unsigned)c, pchunks, pchunk_len, (uint32_t *)phist.p);
        msm_part_scan_kernel<<<(unsigned)(((size_t)nwin * nh + 255) / 256), 256, 0, cur_stream()>>>((uint32_t *)phist.p, nwin, (unsigned)c, pchunks,
                                                                                  (const uint64_t *)starts.p, (uint64_t *)poff.p);
        msm_part_scatter_kernel<<<nwin * pchunks, kPartBlock, 0, cur_stream()>>>((const uint16_t *)digits.p, n, (unsigned)c, pchunks, pchunk_len,
                                                                 (const uint64_t *)poff.p, (uint32_t *)part_e.p, (uint8_t *)part_l.p);
        ZK_HIP(hipFuncSetAttribute((const void *)msm_fine_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFineLdsBytes));
        msm_fine_scatter_kernel<<<nwin * nh, kSortBlock, kFineLdsBytes, cur_stream()>>>((const uint32_t *)part_e.p, (const uint8_t *)part_l.p, (unsigned)c,
                                                           (const uint64_t *)starts.p, (uint32_t *)sorted.p);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipStreamSynchronize(cur_stream()));      // the intermediates are freed on scope exit
    } else {
        msm_scatter_kernel<<<nwin1 * nchunks, kSortBlock, lds_bytes, cur_stream()>>>((const uint16_t *)digits.p, n, (unsigned)c, nchunks, chunk_len,
                                                                        (const uint32_t *)hist.p, (const uint64_t *)starts.p, (uint32_t *)sorted.p,
                                                                        by_chunk);
        ZK_HIP(hipGetLastError());
    }
    }
    ZK_TRY(ev.mark());
    // bucket sums
    ZK_TRY(partials.alloc(((size_t)nseg ? nseg : 1) * kXyzzUBytes));      // XYZZ in the internal form (g1u.cuh) from here on
    size_t red_bytes = ((size_t)nwin << (c - 1)) * kXyzzUBytes;
    ZK_TRY(A.alloc(red_bytes));
    ZK_TRY(R.alloc(red_bytes));
    if (c < 6) ZK_HIP(hipMemsetAsync(R.p, 0, red_bytes, cur_stream()));     // all-zero XYZZ = infinity (ZZ = 0); A (and, c >= 6, R) are written whole by the combine kernel
    if (nseg) {
        ZK_TRY(launch_msm_bucket_sum(d_bases, (const uint32_t *)sorted.p, (const uint64_t *)starts.p, (const uint32_t *)seg_starts.p,
                                     nbuckets, seg_len, entries, partials.p, cur_stream()));
    }
    // heavy buckets (skewed scalars, or the short top window): combine partials 16 at a time until every
    // bucket has at most 16, so no lane ever runs a long serial chain of full additions
    const unsigned kGroup = 16;
    DevBuf lvl_partials[2], lvl_starts[2];
    const void *cur_partials = partials.p;
    const uint32_t *cur_starts = (const uint32_t *)seg_starts.p;
    for (int lvl = 0; max_segs > kGroup; lvl++) {
        DevBuf &np = lvl_partials[lvl & 1], &ns = lvl_starts[lvl & 1];
        ns.release();
        np.release();
        ZK_TRY(ns.alloc((nbuckets + 2) * 4));
        {
            const size_t ntiles = (nbuckets + kScanTile - 1) / kScanTile;
            DevBuf ts, tm;
            ZK_TRY(ts.alloc(ntiles * 4));
            ZK_TRY(tm.alloc(ntiles * 4));
            ZK_TRY(msm_seg_scan(SegFromGroups{cur_starts, kGroup}, nbuckets, (uint32_t *)ts.p, (uint32_t *)tm.p, (uint32_t *)ns.p, cur_stream()));
        }
        ZK_HIP(zk::memcpy_on_stream(tail, (uint32_t *)ns.p + nbuckets, 8, hipMemcpyDeviceToHost));
        ZK_TRY(np.alloc(((size_t)tail[0] ? tail[0] : 1) * kXyzzUBytes));
        ZK_TRY(launch_msm_partials_regroup(cur_partials, cur_starts, (const uint32_t *)ns.p, nbuckets, kGroup, tail[0], np.p, cur_stream()));
        cur_partials = np.p;
        cur_starts = (const uint32_t *)ns.p;
        max_segs = tail[1];
    }
    ZK_TRY(launch_msm_bucket_combine(cur_partials, cur_starts, nwin, (unsigned)c, A.p, c >= 6 ? R.p : nullptr, cur_stream()));
    ZK_TRY(ev.mark());
    std::vector<G1Xyzz> sums(nwin);
    if (c >= 6) {
        // two-stage weighted bucket sum (msm_reduce.hip): slot b = h L + l weighs b + 1 = L h + (l + 1)
        const unsigned cm1 = (unsigned)c - 1, k = cm1 / 2, hb = cm1 - k, mbits = hb > k ? hb : k;
        for (unsigned lvl = 0; lvl < mbits; lvl++) {
            size_t hh = lvl < hb ? ((size_t)1 << (hb - 1 - lvl)) : 0, lh = lvl < k ? ((size_t)1 << (k - 1 - lvl)) : 0;
            ZK_TRY(launch_msm_plain_level(A.p, R.p, nwin, cm1, k, hh, lh, cur_stream()));
        }
        DevBuf X, Y, out3;                                  // X: the arrays C (per window) then D, zero-padded; Y: their R arrays
        const size_t small_bytes = 2 * ((size_t)nwin << mbits) * kXyzzUBytes;
        ZK_TRY(X.alloc(small_bytes));
        ZK_TRY(Y.alloc(small_bytes));
        ZK_TRY(out3.alloc(3 * (size_t)nwin * sizeof(G1Xyzz)));
        ZK_TRY(launch_msm_gather_cd(A.p, R.p, nwin, cm1, k, mbits, X.p, cur_stream()));
        static const bool by_levels = [] { const char *e = getenv("ZK_MSM_WEIGHTED_LEVELS"); return e && e[0] == '1'; }();   // measurement: the r3 form
        if (mbits <= 8 && !by_levels) {                     // by the bits of the weight: one addition per tree level (msm_reduce.hip)
            DevBuf S;
            ZK_TRY(S.alloc((size_t)2 * nwin * (mbits + 1) * kXyzzUBytes));
            ZK_TRY(launch_msm_weighted_bits(X.p, Y.p, 2 * nwin, mbits, S.p, cur_stream()));
        } else {
            ZK_HIP(hipMemsetAsync(Y.p, 0, small_bytes, cur_stream()));
            ZK_TRY(launch_msm_weighted_tail(X.p, Y.p, 2 * nwin, mbits, cur_stream()));   // 2 nwin problems of 2^mbits entries, 0-based weights
        }
        ZK_TRY(launch_msm_two_stage_out(X.p, Y.p, nwin, mbits, out3.p, cur_stream()));
        std::vector<G1Xyzz> o(3 * (size_t)nwin);
        ZK_HIP(hipMemcpyAsync(o.data(), out3.p, o.size() * sizeof(G1Xyzz), hipMemcpyDeviceToHost, cur_stream()));
        ZK_TRY(ev.mark());
        ZK_HIP(hipStreamSynchronize(cur_stream()));
        for (unsigned w = 0; w < nwin; w++) {               // S_w = 2^k sum_h h D[h] + (sum_l l C[l] + sum_l C[l])
            G1Xyzz hi = o[3 * (size_t)w + 2];
            for (unsigned i = 0; i < k; i++) hi = g1_dbl(hi);
            sums[w] = g1_add(hi, g1_add(o[3 * (size_t)w], o[3 * (size_t)w + 1]));
        }
    } else {
        // c - 1 halving levels over the 2^(c-1) slots of every window, in place; window sum = R[0] + A[0]
        for (size_t half = (size_t)1 << (c - 2); half >= 1; half >>= 1) {
            ZK_TRY(launch_msm_reduce_level(A.p, R.p, nwin, (unsigned)c, half, cur_stream()));
            if (half == 1) break;
        }
        DevBuf wsums;
        ZK_TRY(wsums.alloc((size_t)nwin * sizeof(G1Xyzz)));
        ZK_TRY(launch_msm_window_sums(A.p, R.p, nwin, (unsigned)c, wsums.p, cur_stream()));
        ZK_HIP(hipMemcpyAsync(sums.data(), wsums.p, (size_t)nwin * sizeof(G1Xyzz), hipMemcpyDeviceToHost, cur_stream()));
        ZK_TRY(ev.mark());
        ZK_HIP(hipStreamSynchronize(cur_stream()));
    }
    if (shifted) {                                          // the shifts live in the bases: bucket set j IS MSM j
        for (unsigned j = 0; j < batch; j++) result[j] = sums[j];
    } else if (pre) {                                       // likewise: the one bucket set is the MSM
        result[0] = sums[0];
    } else {                                                // window combination (host, W points): acc = 2^c * acc + S_w
        G1Xyzz acc = g1_xyzz_inf();
        for (int w = (int)nwin1 - 1; w >= 0; w--) {
            for (int k = 0; k < c; k++) acc = g1_dbl(acc);
            acc = g1_add(acc, sums[w]);
        }
        result[0] = acc;
    }
    if (stats) {
        stats->window_bits = c;
        stats->windows = (int)nwin1;
        stats->terms = n;
        stats->entries = entries;
        stats->segments = nseg;
        stats->ms_digits = ev.ms(0, 1);
        stats->ms_sort = ev.ms(1, 2);
        stats->ms_buckets = ev.ms(2, 3);
        stats->ms_reduce = ev.ms(3, 4);
        stats->ms_total = ev.ms(0, 4);
    }
    return ZK_OK;
}

// sum_i [s_i] B_i ; result as XYZZ on the host
int msm_device(const void *d_scalars, const zk_g1_bases *bases, size_t n, int c, G1Xyzz *result, zk_msm_stats *stats) {
    if (bases->pre_u && n == bases->n && (c == 0 || c == bases->pre_c))   // window-shifted copies exist: ONE bucket set (zk_g1_bases_precompute)
        return msm_core(d_scalars, bases->pre_u, n, 1, false, bases->pre_c, result, stats, true);
    const void *d_bases = nullptr;
    ZK_TRY(bases_u(bases, &d_bases));                       // pre-converted points (cached on the handle)
    return msm_core(d_scalars, d_bases, n, 1, false, c, result, stats);
}

int bases_alloc(size_t n, zk_g1_bases **out) {
    void *d = nullptr;
    ZK_HIP(hipMalloc(&d, n * sizeof(G1Affine)));
    *out = new zk_g1_bases{n, d, nullptr};
    return ZK_OK;
}
int bases_u(const zk_g1_bases *b, const void **out) {
    zk_g1_bases *mb = const_cast<zk_g1_bases *>(b);          // a cache of the same immutable points
    if (!mb->dptr_u) {
        void *d = nullptr;
        ZK_HIP(hipMalloc(&d, b->n * kBaseUBytes));
        int rc = launch_g1_bases_to_u(b->dptr, b->n, d, cur_stream());
        if (rc != ZK_OK) { (void)hipFree(d); return rc; }
        mb->dptr_u = d;
    }
    *out = mb->dptr_u;
    return ZK_OK;
}

// XYZZ (device) -> affine bases (device)
int normalize_to_bases(const void *d_xyzz, size_t n, zk_g1_bases **out) {
    ZK_TRY(bases_alloc(n, out));
    ZK_TRY(launch_batch_to_affine(d_xyzz, n, (*out)->dptr, cur_stream()));
    ZK_HIP(hipStreamSynchronize(cur_stream()));
    return ZK_OK;
}

// window table of the generator for the fixed-base kernel (g1_setup.hip), built once per device: the byte-window table
// table8[j * 256 + v] = [v * 256^j] G on the host (8 K points), its 16-bit-window form on the device
std::mutex g_tab_mu;
std::vector<void *> g_gen_table;
int generator_table(const void **out) {
    int dev = 0;
    ZK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_tab_mu);
    if ((int)g_gen_table.size() <= dev) g_gen_table.resize(dev + 1, nullptr);
    if (!g_gen_table[dev]) {
        std::vector<G1Xyzz> tab(32 * 256);
        G1Xyzz base = g1_from_affine(g1_generator());
        for (int j = 0; j < 32; j++) {
            tab[j * 256] = g1_xyzz_inf();
            for (int v = 1; v < 256; v++) tab[j * 256 + v] = g1_add(tab[j * 256 + v - 1], base);
            for (int k = 0; k < 8; k++) base = g1_dbl(base);
        }
        // batch normalisation on the host: one inversion for the whole table
        std::vector<FqE> prefix(tab.size());
        FqE run = fe_one<Fq>();
        for (size_t i = 0; i < tab.size(); i++) {
            prefix[i] = run;
            if (!g1_is_inf(tab[i])) run = fe_mul<Fq>(run, fe_mul<Fq>(tab[i].zz, tab[i].zzz));
        }
        FqE inv = fe_inv<Fq>(run);
        std::vector<G1Affine> aff(tab.size());
        for (size_t i = tab.size(); i-- > 0;) {
            if (g1_is_inf(tab[i])) { aff[i].x = fe_zero<Fq>(); aff[i].y = fe_zero<Fq>(); continue; }
            FqE t = fe_mul<Fq>(inv, prefix[i]);
            inv = fe_mul<Fq>(inv, fe_mul<Fq>(tab[i].zz, tab[i].zzz));
            aff[i].x = fe_mul<Fq>(tab[i].x, fe_mul<Fq>(t, tab[i].zzz));
            aff[i].y = fe_mul<Fq>(tab[i].y, fe_mul<Fq>(t, tab[i].zz));
        }
        // 16-bit windows on the device: pairs of byte entries added, normalised, converted to the internal form (once per device)
        const size_t n16 = (size_t)16 * 65536;
        DevBuf t8, xyzz, aff16;
        ZK_TRY(t8.alloc(aff.size() * sizeof(G1Affine)));
        ZK_HIP(zk::memcpy_on_stream(t8.p, aff.data(), aff.size() * sizeof(G1Affine), hipMemcpyHostToDevice));
        ZK_TRY(xyzz.alloc(n16 * sizeof(G1Xyzz)));
        ZK_TRY(aff16.alloc(n16 * sizeof(G1Affine)));
        void *d = nullptr;
        ZK_HIP(hipMalloc(&d, n16 * kBaseUBytes));
        int rc = launch_fixed_table16(t8.p, xyzz.p, cur_stream());
        if (rc == ZK_OK) rc = launch_batch_to_affine(xyzz.p, n16, aff16.p, cur_stream());
        if (rc == ZK_OK) rc = launch_g1_bases_to_u(aff16.p, n16, d, cur_stream());
        if (rc == ZK_OK && hipStreamSynchronize(cur_stream()) != hipSuccess) rc = ZK_E_HIP;
        if (rc != ZK_OK) { (void)hipFree(d); return rc; }
        g_gen_table[dev] = d;
    }
    *out = g_gen_table[dev];
    return ZK_OK;
}

int lagrange_basis_device(const uint64_t *taus, size_t ntaus, zk_table **out) {
    if (!taus || ntaus == 0 || ntaus > 40) return ZK_E_ARG;       // "requires at least one variable" trusted_setup.rs:26
    size_t n = (size_t)1 << ntaus;
    zk_table *t = nullptr;
    ZK_TRY(zk_table_alloc(ZK_FR381, n, &t));
    EqBuilder<Fr381> eb;                                          // variable 0 = MSB (:36); outer products of half tables
    int rc = eb.build(taus, (uint32_t)ntaus, t->dptr);
    if (rc == ZK_OK && hipStreamSynchronize(cur_stream()) != hipSuccess) rc = ZK_E_HIP;
    if (rc != ZK_OK) { zk_table_free(t); return rc; }
    *out = t;
    return ZK_OK;
}

}  // namespace

extern "C" {

int zk_g1_bases_upload(const uint64_t *affine, size_t n, zk_g1_bases **out) {
    if (!affine || !out || n == 0) return ZK_E_ARG;
    ZK_TRY(require_device());
    ZK_TRY(bases_alloc(n, out));
    hipError_t e = zk::memcpy_on_stream((*out)->dptr, affine, n * sizeof(G1Affine), hipMemcpyHostToDevice);
    if (e != hipSuccess) { zk_g1_bases_free(*out); *out = nullptr; ZK_HIP(e); }
    return ZK_OK;
}
int zk_g1_bases_download(const zk_g1_bases *b, uint64_t *affine) {
    if (!b || !affine) return ZK_E_ARG;
    ZK_HIP(zk::memcpy_on_stream(affine, b->dptr, b->n * sizeof(G1Affine), hipMemcpyDeviceToHost));
    return ZK_OK;
}
int zk_g1_bases_free(zk_g1_bases *b) {
    if (!b) return ZK_OK;
    if (b->dptr) ZK_HIP(hipFree(b->dptr));
    if (b->dptr_u) ZK_HIP(hipFree(b->dptr_u));
    if (b->pre_u) ZK_HIP(hipFree(b->pre_u));
    if (b->own_key) zk_kzg_opening_key_free(b->own_key);
    delete b;
    return ZK_OK;
}
size_t zk_g1_bases_len(const zk_g1_bases *b) { return b ? b->n : 0; }

int zk_g1_generator(uint64_t *out12) {
    if (!out12) return ZK_E_ARG;
    affine_to_u64(g1_generator(), out12);
    return ZK_OK;
}
int zk_g1_is_on_curve(const uint64_t *p12) {
    if (!p12) return ZK_E_ARG;
    return g1_on_curve(affine_from_u64(p12)) ? 1 : 0;
}

int zk_g1_add(const uint64_t *p12, const uint64_t *q12, uint64_t *out12) {
    if (!p12 || !q12 || !out12) return ZK_E_ARG;
    affine_to_u64(g1_to_affine(g1_madd(g1_from_affine(affine_from_u64(p12)), affine_from_u64(q12))), out12);
    return ZK_OK;
}
int zk_g1_mul_fr(const uint64_t *p12, const uint64_t *scalar_fr, uint64_t *out12) {
    if (!p12 || !scalar_fr || !out12) return ZK_E_ARG;
    Fe<Fr381> k = fe_to_canonical<Fr381>(load_el<Fr381>(scalar_fr));
    affine_to_u64(g1_to_affine(g1_mul_canonical(affine_from_u64(p12), k.l, 8)), out12);
    return ZK_OK;
}

int zk_g1_bases_synthetic(size_t n, const uint64_t *a_fr, const uint64_t *d_fr, zk_g1_bases **out) {
    if (!a_fr || !d_fr || !out || n == 0) return ZK_E_ARG;
    ZK_TRY(require_device());
    Fe<Fr381> a = fe_to_canonical<Fr381>(load_el<Fr381>(a_fr)), d = fe_to_canonical<Fr381>(load_el<Fr381>(d_fr));
    G1Affine g = g1_generator();
    G1Affine dstep = g1_to_affine(g1_mul_canonical(g, d.l, 8));
    DevBuf xyzz;
    ZK_TRY(xyzz.alloc(n * sizeof(G1Xyzz)));
    ZK_TRY(launch_synthetic_bases(g, dstep, a, d, n, 64, xyzz.p, cur_stream()));
    return normalize_to_bases(xyzz.p, n, out);
}

int zk_msm_g1(const zk_table *scalars, const zk_g1_bases *bases, int window_bits, uint64_t *out12, zk_msm_stats *stats) {
    if (!scalars || !bases || !out12) return ZK_E_ARG;
    if (scalars->field != ZK_FR381) return ZK_E_ARG;
    if (scalars->len != bases->n) return ZK_E_KZG_LEN;
    ZK_TRY(require_device());
    G1Xyzz r;
    ZK_TRY(msm_device(scalars->dptr, bases, scalars->len, window_bits, &r, stats));
    affine_to_u64(g1_to_affine(r), out12);
    return ZK_OK;
}

// One pre-converted copy of the points per window, [w][i] = 2^(c w) B_i: every window of a later MSM on these bases then feeds ONE
// bucket set -- W n bucket additions into 2^(c-1) buckets, ONE bucket reduction, no window combination -- which is what lets the window
// grow to 22 bits (12 additions per term instead of 16).  Paid once per setup: (W - 1) c doublings per point, W x 128 bytes per point.
int zk_g1_bases_precompute(zk_g1_bases *b, int window_bits) {
    if (!b) return ZK_E_ARG;
    ZK_TRY(require_device());
    int c = window_bits;
    if (c == 0) {                                        // W n + 2.8 x 2^(c-1) addition-equivalents: 22 bits from 2^22 points up
        const unsigned lg = ilog2(b->n);
        c = lg >= 22 ? 22 : lg >= 18 ? 20 : lg >= 14 ? 16 : 13;
    }
    if (c < 9 || c > 24) return ZK_E_ARG;
    const unsigned nwin = (256 + c - 1) / c;
    if ((size_t)nwin * b->n >= ((size_t)1 << 31)) return ZK_E_ARG;
    if (b->pre_u && b->pre_c == c) return ZK_OK;
    if (b->pre_u) { ZK_HIP(hipFree(b->pre_u)); b->pre_u = nullptr; b->pre_c = 0; }
    const size_t n = b->n;
    void *tab = nullptr, *x0 = nullptr, *x1 = nullptr, *aff = nullptr;
    ZK_HIP(hipMalloc(&tab, (size_t)nwin * n * kBaseUBytes));
    struct Guard { void **p; ~Guard() { if (*p) (void)hipFree(*p); } } g0{&tab}, g1{&x0}, g2{&x1}, g3{&aff};
    ZK_HIP(hipMalloc(&x0, n * sizeof(G1Xyzz)));
    ZK_HIP(hipMalloc(&x1, n * sizeof(G1Xyzz)));
    ZK_HIP(hipMalloc(&aff, n * sizeof(G1Affine)));
    ZK_TRY(launch_g1_bases_to_u(b->dptr, n, tab, cur_stream()));                       // window 0: the points themselves
    const void *src = b->dptr;
    void *dst = x0, *other = x1;
    for (unsigned w = 1; w < nwin; w++) {                                             // window w = c doublings of window w - 1
        ZK_TRY(launch_g1_shift(src, w == 1 ? 0 : 1, n, (unsigned)c, dst, cur_stream()));
        ZK_TRY(launch_batch_to_affine(dst, n, aff, cur_stream()));
        ZK_TRY(launch_g1_bases_to_u(aff, n, (char *)tab + (size_t)w * n * kBaseUBytes, cur_stream()));
        src = dst;
        void *t = dst; dst = other; other = t;
    }
    ZK_HIP(hipStreamSynchronize(cur_stream()));
    b->pre_u = tab;
    b->pre_c = c;
    b->pre_nwin = nwin;
    tab = nullptr;                                       // kept
    return ZK_OK;
}
int zk_g1_bases_precomputed_window(const zk_g1_bases *b) { return b ? b->pre_c : 0; }

int zk_kzg_lagrange_basis(const uint64_t *taus, size_t ntaus, zk_table **out) {
    if (!out) return ZK_E_ARG;
    ZK_TRY(require_device());
    return lagrange_basis_device(taus, ntaus, out);
}

int zk_kzg_setup_g1(const uint64_t *taus, size_t ntaus, zk_g1_bases **out) {
    if (!out) return ZK_E_ARG;
    ZK_TRY(require_device());
    zk_table *basis = nullptr;
    ZK_TRY(lagrange_basis_device(taus, ntaus, &basis));              // trusted_setup.rs:12
    size_t n = basis->len;
    const void *table = nullptr;
    DevBuf xyzz;
    int rc = generator_table(&table);
    if (rc == ZK_OK) rc = xyzz.alloc(n * sizeof(G1Xyzz));
    if (rc == ZK_OK) {
        rc = launch_fixed_base_mul(basis->dptr, n, table, xyzz.p, cur_stream());   // :51-60
    }
    if (rc == ZK_OK) rc = normalize_to_bases(xyzz.p, n, out);
    zk_table_free(basis);
    return rc;
}

int zk_kzg_commit(const zk_table *poly, const zk_g1_bases *g1_powers, uint64_t *out12) {
    if (!poly || !g1_powers || !out12) return ZK_E_ARG;
    if (poly->field != ZK_FR381) return ZK_E_ARG;
    if (poly->len != g1_powers->n) return ZK_E_KZG_LEN;              // multilinear_kzg.rs:29-33
    return zk_msm_g1(poly, g1_powers, 0, out12, nullptr);
}

int zk_kzg_opening_key_new(const zk_g1_bases *g1, zk_kzg_opening_key **out) {
    if (!g1 || !out) return ZK_E_ARG;
    if (!is_pow2(g1->n)) return ZK_E_NOT_POW2;
    ZK_TRY(require_device());
    std::unique_ptr<zk_kzg_opening_key> key(new zk_kzg_opening_key());
    key->nvars = ilog2(g1->n);
    key->level.assign(key->nvars + 1, nullptr);
    const zk_g1_bases *cur = g1;
    int rc = ZK_OK;
    // levels of more than kChainLen points: pair sums of the previous (affine) level, normalised one by one.  From there on the
    // sums stay XYZZ, level after level end to end in one buffer, and are normalised together: a normalisation is one ~1.3 ms
    // inversion chain whatever its size, and there are up to 13 such levels.
    const size_t kChainLen = 4096;
    size_t t = 1;
    for (; t <= key->nvars && rc == ZK_OK && cur->n / 2 > kChainLen; t++) {
        size_t half = cur->n / 2;
        DevBuf xyzz;
        rc = xyzz.alloc(half * sizeof(G1Xyzz));
        if (rc != ZK_OK) break;
        rc = launch_g1_pair_add(cur->dptr, half, xyzz.p, cur_stream());
        if (rc != ZK_OK) break;
        rc = normalize_to_bases(xyzz.p, half, &key->level[t]);
        cur = key->level[t];
    }
    if (rc == ZK_OK && t <= key->nvars) {
        const size_t first = cur->n / 2, total = 2 * first - 1;         // sizes first, first / 2, ..., 1
        DevBuf chain, aff;
        rc = chain.alloc(total * sizeof(G1Xyzz));
        if (rc == ZK_OK) rc = aff.alloc(total * sizeof(G1Affine));
        if (rc == ZK_OK) rc = launch_g1_pair_add(cur->dptr, first, chain.p, cur_stream());
        size_t off = 0;
        for (size_t half = first / 2; half >= 1 && rc == ZK_OK; half /= 2) {
            rc = launch_g1_pair_add_xyzz((const char *)chain.p + off * sizeof(G1Xyzz), half, (char *)chain.p + (off + 2 * half) * sizeof(G1Xyzz), cur_stream());
            off += 2 * half;
        }
        if (rc == ZK_OK) rc = launch_batch_to_affine(chain.p, total, aff.p, cur_stream());
        off = 0;
        for (size_t len = first; len >= 1 && rc == ZK_OK; len /= 2, t++) {
            rc = bases_alloc(len, &key->level[t]);
            if (rc == ZK_OK && hipMemcpyAsync(key->level[t]->dptr, (const char *)aff.p + off * sizeof(G1Affine), len * sizeof(G1Affine), hipMemcpyDeviceToDevice,
                                              cur_stream()) != hipSuccess) rc = ZK_E_HIP;
            off += len;
        }
        if (rc == ZK_OK && hipStreamSynchronize(cur_stream()) != hipSuccess) rc = ZK_E_HIP;
    }
    if (rc == ZK_OK && key->nvars >= 2) {                   // with one variable there is a single 1-term MSM: nothing to batch
        key->small_bits = (unsigned)(key->nvars - 1 < open_batch_bits() ? key->nvars - 1 : open_batch_bits());
        key->small_t0 = key->nvars - key->small_bits;
        const size_t nlev = key->small_bits + 1, total = ((size_t)1 << nlev) - 1;
        key->small_c = pick_window((size_t)1 << key->small_bits);
        const unsigned c = (unsigned)key->small_c, nwin = (256 + c - 1) / c;
        DevBuf aff, xall, affall;                           // the levels end to end; their shifted copies 2^(c w) B, w = 1 .. nwin - 1
        rc = aff.alloc(total * sizeof(G1Affine));
        if (rc == ZK_OK) rc = xall.alloc((size_t)(nwin - 1) * total * sizeof(G1Xyzz));
        if (rc == ZK_OK) rc = affall.alloc((size_t)(nwin - 1) * total * sizeof(G1Affine));
        hipError_t e = hipSuccess;
        if (rc == ZK_OK) e = hipMalloc(&key->small_u, (size_t)nwin * total * kBaseUBytes);
        for (size_t j = 0; j < nlev && rc == ZK_OK && e == hipSuccess; j++) {
            const zk_g1_bases *lv = key->level[key->small_t0 + j];
            const size_t off = ((size_t)1 << nlev) - ((size_t)1 << (nlev - j));
            e = hipMemcpyAsync((char *)aff.p + off * sizeof(G1Affine), lv->dptr, lv->n * sizeof(G1Affine), hipMemcpyDeviceToDevice, cur_stream());
        }
        if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); rc = ZK_E_HIP; }
        // window 0 as it is; every further window is c doublings of the previous one (XYZZ to XYZZ), and all of them are
        // normalised in ONE batch (r1: one normalisation per window cost its ~1.3 ms inversion latency 15 times)
        if (rc == ZK_OK) rc = launch_g1_bases_to_u(aff.p, total, key->small_u, cur_stream());
        for (unsigned w = 1; w < nwin && rc == ZK_OK; w++) {
            const void *src = w == 1 ? aff.p : (const void *)((const char *)xall.p + (size_t)(w - 2) * total * sizeof(G1Xyzz));
            rc = launch_g1_shift(src, w == 1 ? 0 : 1, total, c, (char *)xall.p + (size_t)(w - 1) * total * sizeof(G1Xyzz), cur_stream());
        }
        if (rc == ZK_OK && nwin > 1) rc = launch_batch_to_affine(xall.p, (size_t)(nwin - 1) * total, affall.p, cur_stream());
        if (rc == ZK_OK && nwin > 1) rc = launch_g1_bases_to_u(affall.p, (size_t)(nwin - 1) * total, (char *)key->small_u + total * kBaseUBytes, cur_stream());
        if (rc == ZK_OK && hipStreamSynchronize(cur_stream()) != hipSuccess) rc = ZK_E_HIP;
    }
    if (rc != ZK_OK) { zk_kzg_opening_key_free(key.release()); return rc; }
    *out = key.release();
    return ZK_OK;
}
int zk_kzg_opening_key_precompute(zk_kzg_opening_key *k, int window_bits, size_t min_points) {
    if (!k) return ZK_E_ARG;
    if (min_points == 0) min_points = (size_t)1 << 18;
    const size_t nbig = k->small_u ? k->small_t0 - 1 : k->nvars;     // levels 1 .. nbig take a plain MSM each (the rest go through the batched pass)
    for (size_t t = 1; t <= nbig; t++)
        if (k->level[t] && k->level[t]->n >= min_points) ZK_TRY(zk_g1_bases_precompute(k->level[t], window_bits));
    return ZK_OK;
}
// sum of all the bases of the key's setup = its last pre-summed level (one point), affine
extern "C++" int zk::kzg_key_total(const zk_kzg_opening_key *k, const zk_g1_bases *g1, uint64_t *out12) {
    const zk_g1_bases *lv = (k && k->nvars >= 1) ? k->level[k->nvars] : g1;
    if (!lv || lv->n != 1) return ZK_E_ARG;
    return zk_g1_bases_download(lv, out12);
}
int zk_kzg_opening_key_free(zk_kzg_opening_key *k) {
    if (!k) return ZK_OK;
    if (k->small_u) (void)hipFree(k->small_u);
    for (zk_g1_bases *b : k->level) zk_g1_bases_free(b);
    delete k;
    return ZK_OK;
}

// Side streams for the level MSMs of an opening (one set per device, created on first use and kept: the scratch pool caches
// its blocks per stream).
static std::mutex g_open_mu;
static std::vector<std::vector<hipStream_t>> g_open_streams;
static int open_side_streams(int dev, unsigned count, std::vector<hipStream_t> &out) {
    std::lock_guard<std::mutex> lk(g_open_mu);
    if ((int)g_open_streams.size() <= dev) g_open_streams.resize(dev + 1);
    std::vector<hipStream_t> &v = g_open_streams[dev];
    while (v.size() < count) {
        hipStream_t st = nullptr;
        ZK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        v.push_back(st);
    }
    out.assign(v.begin(), v.begin() + count);
    return ZK_OK;
}
static unsigned open_threads() {
    static const unsigned k = [] { const char *e = getenv("ZK_KZG_OPEN_THREADS"); int v = e ? atoi(e) : 3; return (unsigned)(v < 1 ? 1 : (v > 4 ? 4 : v)); }();
    return k;
}

// open_and_prove (multilinear_kzg.rs:50-126).  The quotients of all levels come from one chain of folds on the caller's stream;
// the level MSMs are independent of each other, so the large ones and the batched pass of the small ones are then handed to
// `ZK_KZG_OPEN_THREADS` (default 3) host threads, each driving its own stream: the latency-bound phases of one MSM (bucket
// reduction levels, scans, host read-backs: ~1.4 ms per MSM) run under the bucket phase of another.
int zk_kzg_open(const zk_table *poly, const zk_g1_bases *g1_powers, const zk_kzg_opening_key *key, const uint64_t *opening,
                size_t nopen, size_t n_g2, uint64_t *evaluation, uint64_t *proofs) {
    if (!poly || !g1_powers || !opening || !evaluation || !proofs) return ZK_E_ARG;
    if (nopen != n_g2) return ZK_E_KZG_LEN;                          // :60-64
    return zk::kzg_open_core(poly, g1_powers, key, opening, nopen, nullptr, evaluation, proofs, nullptr);
}

// The body of open_and_prove.  `v_given` (may be null): the value to subtract instead of poly(opening) -- a rank of a sharded
// opening passes the GLOBAL evaluation (its shard's own evaluation is something else), the replicated tail passes zero (its table is
// already f - v).  `last` (may be null): the single entry left of f - v after all the folds.
extern "C++" int zk::kzg_open_core(const zk_table *poly, const zk_g1_bases *g1_powers, const zk_kzg_opening_key *key, const uint64_t *opening,
                      size_t nopen, const uint64_t *v_given, uint64_t *evaluation, uint64_t *proofs, uint64_t *last) {
    if (!poly || !g1_powers || (!opening && nopen) || !evaluation || (!proofs && nopen)) return ZK_E_ARG;
    if (poly->field != ZK_FR381) return ZK_E_ARG;
    if (!is_pow2(poly->len)) return ZK_E_NOT_POW2;
    if (ilog2(poly->len) != nopen) return ZK_E_KZG_LEN;              // :55-59
    if (poly->len != g1_powers->n) return ZK_E_KZG_LEN;              // the zip of :100-103 needs equal lengths (asserted at commit :29)
    ZK_TRY(require_device());
    zk_kzg_opening_key *own = nullptr;
    if (!key) {
        ZK_TRY(zk_kzg_opening_key_new(g1_powers, &own));
        key = own;
    }
    int rc = ZK_OK;
    if (key->nvars != nopen) rc = ZK_E_KZG_LEN;
    zk_table *sub = nullptr, *nxt = nullptr;
    if (rc == ZK_OK && v_given) memcpy(evaluation, v_given, 32);
    if (rc == ZK_OK && !v_given) rc = zk_mle_evaluate(poly, opening, nopen, evaluation);        // :70
    if (rc == ZK_OK) rc = table_alloc_pooled(ZK_FR381, poly->len, &sub);                        // pooled: a small opening is not two hipMalloc / hipFree pairs
    if (rc == ZK_OK) rc = zk_mle_sub_scalar(poly, evaluation, sub, nullptr);                    // :74-80
    if (rc == ZK_OK && poly->len >= 2) rc = table_alloc_pooled(ZK_FR381, poly->len / 2, &nxt);
    // quotients of the batched (small) levels are collected, zero-padded, in one scalar buffer [j][2^small_bits];
    // those of the large levels lie end to end in `bigq` (level i at offset len - len / 2^i)
    const bool batched = key->small_u != nullptr && rc == ZK_OK;
    const size_t nlev = key->small_bits + 1, small_total = ((size_t)1 << nlev) - 1;
    const size_t nbig = batched ? key->small_t0 - 1 : nopen;          // levels t = 1 .. nbig take a plain MSM each
    DevBuf smallq, bigq;
    if (batched) rc = smallq.alloc(small_total * 32);
    if (rc == ZK_OK && nbig) rc = bigq.alloc(poly->len * 32);
    std::vector<size_t> big_off(nbig + 1, 0);
    for (size_t i = 0; i < nopen && rc == ZK_OK; i++) {                                          // :86
        size_t half = sub->len / 2;
        const size_t t = i + 1;
        const bool small = t > nbig;
        // batched level j = t - small_t0 sits at offset 2^nlev - 2^(nlev - j) of the end-to-end scalar buffer
        void *qdst = small ? (void *)((char *)smallq.p + (((size_t)1 << nlev) - ((size_t)1 << (nlev - (t - key->small_t0)))) * 32)
                           : (void *)((char *)bigq.p + big_off[i] * 32);
        if (!small) big_off[i + 1] = big_off[i] + half;
        // quotient = hi half - lo half (compute_quotient_polynomial :165-179)
        elementwise_kernel<Fr381, OP_HI_MINUS_LO><<<grid_for(half), kBlock, 0, cur_stream()>>>(sub->dptr, nullptr, qdst, half, fe_zero<Fr381>());
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); rc = ZK_E_HIP; break; }
        nxt->len = half;
        rc = zk_mle_fold(sub, 0, opening + 4 * i, nxt, nullptr);                                 // :113-119
        zk_table *tt = sub; sub = nxt; nxt = tt;
    }
    if (rc == ZK_OK && hipStreamSynchronize(cur_stream()) != hipSuccess) rc = ZK_E_HIP;
    if (rc == ZK_OK && last && zk_table_download(sub, last) != ZK_OK) rc = ZK_E_HIP;               // `sub` has one entry left
    // proof_i = sum_j [blown_up(q)[j]] B_j  (:96-107)  ==  sum_k [q[k]] B^(i+1)_k : task i < nbig; task nbig = the batched small levels
    const size_t ntasks = nbig + (batched ? 1 : 0);
    auto run_task = [&](size_t k) -> int {
        if (k < nbig) {
            G1Xyzz pi;
            ZK_TRY(msm_device((const char *)bigq.p + big_off[k] * 32, key->level[k + 1], big_off[k + 1] - big_off[k], 0, &pi, nullptr));
            affine_to_u64(g1_to_affine(pi), proofs + 12 * k);
            return ZK_OK;
        }
        std::vector<G1Xyzz> pis(nlev);
        ZK_TRY(msm_core(smallq.p, key->small_u, (size_t)1 << key->small_bits, (unsigned)nlev, true, key->small_c, pis.data(), nullptr));
        for (size_t j = 0; j < nlev; j++) affine_to_u64(g1_to_affine(pis[j]), proofs + 12 * (key->small_t0 + j - 1));
        return ZK_OK;
    };
    const unsigned nthreads = (unsigned)(ntasks < open_threads() ? ntasks : open_threads());
    if (rc == ZK_OK && nthreads <= 1) {
        for (size_t k = 0; k < ntasks && rc == ZK_OK; k++) rc = run_task(k);
    } else if (rc == ZK_OK) {
        int dev = 0;
        std::vector<hipStream_t> streams;
        if (hipGetDevice(&dev) != hipSuccess) rc = ZK_E_HIP;
        if (rc == ZK_OK) rc = open_side_streams(dev, nthreads, streams);
        std::atomic<size_t> next{0};
        std::vector<int> rcs(nthreads, ZK_OK);
        std::vector<std::string> errs(nthreads);
        std::vector<std::thread> workers;
        for (unsigned w = 0; w < nthreads && rc == ZK_OK; w++)
            workers.emplace_back([&, w] {
                if (hipSetDevice(dev) != hipSuccess) { rcs[w] = ZK_E_HIP; return; }
                zk_set_stream((void *)streams[w]);
                for (size_t k; (k = next.fetch_add(1)) < ntasks;) {           // largest MSM first (tasks are in level order)
                    int r = run_task(k);
                    if (r != ZK_OK) { rcs[w] = r; errs[w] = zk_last_error(); break; }
                }
            });
        for (std::thread &th : workers) th.join();
        for (unsigned w = 0; w < nthreads && rc == ZK_OK; w++)
            if (rcs[w] != ZK_OK) { rc = rcs[w]; set_last_error(errs[w]); }
    }
    zk_table_free(sub);
    zk_table_free(nxt);
    zk_kzg_opening_key_free(own);
    return rc;
}

int zk_gkr_prove_succinct(const zk_gate *gates, const size_t *gate_counts, size_t nlayers, const uint64_t *inputs, size_t ninputs,
                          const zk_g1_bases *g1_powers, size_t n_g2, uint64_t *circuit_output, size_t *output_len, uint64_t *claimed_sum,
                          uint64_t *layer_claims, uint64_t *coeffs, uint64_t *challenges, uint64_t *wb_evals, uint64_t *wc_evals,
                          uint64_t *commitment12, uint64_t *rb_evaluation, uint64_t *rb_proofs, uint64_t *rc_evaluation,
                          uint64_t *rc_proofs) {
    if (!g1_powers || !commitment12 || !rb_evaluation || !rb_proofs || !rc_evaluation || !rc_proofs || nlayers == 0) return ZK_E_ARG;
    zk_table *in = nullptr;
    if (!inputs) return ZK_E_ARG;
    if (!is_pow2(ninputs)) return ZK_E_NOT_POW2;                                 // MultilinearPolynomial::new(inputs) :41
    ZK_TRY(table_alloc_pooled(ZK_FR381, ninputs, &in));
    if (zk::memcpy_on_stream(in->dptr, inputs, ninputs * 32, hipMemcpyHostToDevice) != hipSuccess) { zk_table_free(in); return ZK_E_HIP; }
    int rc = zk_kzg_commit(in, g1_powers, commitment12);                         // :42-44
    // the layer loop is gkr_protocol::prove's (same transcript schedule; rb / rc are taken from
    // every layer's challenges here, :121-126, which only matters for the last layer's values)
    if (rc == ZK_OK)
        rc = zk_gkr_prove(ZK_FR381, gates, gate_counts, nlayers, inputs, ninputs, circuit_output, output_len, claimed_sum, layer_claims,
                          coeffs, challenges, wb_evals, wc_evals);
    if (rc == ZK_OK) {
        size_t off = 0;
        for (size_t L = 0; L + 1 < nlayers; L++) off += zk_gkr_rounds(L);
        size_t rounds = zk_gkr_rounds(nlayers - 1), mid = rounds / 2;
        const uint64_t *rb = challenges + off * 4, *rcv = challenges + (off + mid) * 4;
        zk_g1_bases *mb = const_cast<zk_g1_bases *>(g1_powers);                  // a cache of sums of the same immutable points (as bases_u)
        static std::mutex key_mu;
        {
            std::lock_guard<std::mutex> lk(key_mu);
            rc = in->len != g1_powers->n ? ZK_E_KZG_LEN : (mb->own_key ? ZK_OK : zk_kzg_opening_key_new(g1_powers, &mb->own_key));
        }
        const zk_kzg_opening_key *key = mb->own_key;
        if (rc == ZK_OK) rc = zk_kzg_open(in, g1_powers, key, rb, mid, n_g2, rb_evaluation, rb_proofs);            // :154-155
        if (rc == ZK_OK) rc = zk_kzg_open(in, g1_powers, key, rcv, rounds - mid, n_g2, rc_evaluation, rc_proofs);  // :156-157
    }
    zk_table_free(in);
    return rc;
}

}  // extern "C"
