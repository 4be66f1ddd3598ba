// msm_sort_wide.cuh -- the counting sort of Pippenger's (term, window) entries for WIDE windows (c = 17 .. 24 bits).
//
// With c <= 16 a whole window's 2^(c-1) bucket counters fit one workgroup's LDS (msm_kernels.cuh).  Wider windows cut the number
// of bucket additions (W = ceil(256 / c) per term: 16 -> 13 at c = 20) but have 2^16 .. 2^23 buckets per set, so the entries are
// sorted most-significant-digit first in up to three levels of at most 256 bins, each level a chunked, tile-sorted scatter
// (a tile of the chunk is counting-sorted in LDS so that every bin's share leaves as ONE contiguous run -- scattered 4-byte
// stores are L2-transaction bound, measured r1), and no level ever needs a histogram wider than its own bins:
//   level 1  bins = top 7 bits of the bucket id, per window (or, with ONE bucket set fed by every window -- precomputed
//            window-shifted bases -- over all windows); its histogram is taken by the digit kernel itself;
//   level 2  one part of level 1 at a time, `C2` workgroups per part: the middle bits (skipped when <= 8 bits are left);
//   level 3  one workgroup per group of <= 256 buckets: the group usually fits one LDS tile; writes the final order and the
//            per-bucket counts, from which the existing scans (msm_kernels.cuh) derive bucket and segment starts.
// Entry = index of the point (term i; or w * n + i into the window-shifted copies) | sign << 31.  Keys shrink level by level
// (u32 digit -> u16 -> u8).  Nothing here touches a curve point.
#pragma once
#include "msm_kernels.cuh"

namespace zk {

constexpr unsigned kWideTopBits = 7;          // level-1 bins per bucket set
constexpr int kWideTile = 8192;               // elements per LDS tile of the level-1 / level-2 scatters
constexpr int kWideBlock = 1024;
constexpr int kWideFineTile = 8192;           // level 3: a whole group of buckets per tile (44 KB of LDS: three workgroups per CU)

struct WidePlan {
    unsigned c, cb, nwin, hb1, kb1, mb, lb3;  // cb = c - 1 bucket bits = hb1 + mb + lb3; kb1 = cb - hb1 bits left after level 1
    unsigned one_set;                         // every window feeds the same bucket set (entries index the shifted copies)
    unsigned nchunks1;                        // chunks of terms (level 1)
    size_t chunk_len1;
    unsigned nbins1, ctot1;                   // level-1 bins (parent-major) and chunks per bin
    unsigned nb2, c2;                         // level-2 bins per part and workgroups per part (mb > 0)
    size_t ngroups;                           // level-3 groups
    size_t nbuckets;
};

inline WidePlan wide_plan(size_t n, unsigned c, unsigned nwin, bool one_set) {
    WidePlan p{};
    p.c = c; p.cb = c - 1; p.nwin = nwin; p.one_set = one_set ? 1u : 0u;
    p.hb1 = p.cb < kWideTopBits ? p.cb : kWideTopBits;
    p.kb1 = p.cb - p.hb1;
    // level 3 takes groups of 2^lb3 = 256 buckets (fewer only when fewer bits are left): sparse buckets make a group fit ONE LDS tile
    // (one pass), heavy buckets make it several tiles of >= 32 entries per bucket (count first, then tile by tile); in between, one more
    // middle bit halves the groups into the one-tile case
    const size_t all_entries = (size_t)n * nwin;
    p.mb = p.kb1 > 8 ? p.kb1 - 8 : 0;
    p.lb3 = p.kb1 - p.mb;
    const size_t avg_group = (all_entries >> (p.hb1 + p.mb)) / (one_set ? 1u : nwin);
    if (avg_group > (size_t)kWideFineTile * 3 / 4 && avg_group <= (size_t)2 * kWideFineTile && p.lb3 == 8 && p.mb < 8) { p.mb++; p.lb3--; }
    p.chunk_len1 = (n + 255) / 256;
    if (p.chunk_len1 < (size_t)kWideTile) p.chunk_len1 = kWideTile;
    p.nchunks1 = (unsigned)((n + p.chunk_len1 - 1) / p.chunk_len1);
    p.nbins1 = (one_set ? 1u : nwin) << p.hb1;
    p.ctot1 = one_set ? nwin * p.nchunks1 : p.nchunks1;
    p.nb2 = 1u << p.mb;
    const size_t avg_part = ((size_t)n * nwin) / p.nbins1;
    size_t c2 = (avg_part + 32767) / 32768;
    p.c2 = (unsigned)(c2 < 1 ? 1 : (c2 > 64 ? 64 : c2));
    p.ngroups = (size_t)p.nbins1 << p.mb;
    p.nbuckets = (size_t)(one_set ? 1u : nwin) << p.cb;
    return p;
}

// exclusive prefix of cnt[0 .. nbins) (nbins <= 256) into binstart, by the first four waves; every thread of the block calls it
// (blockDim >= 256), `cnt` complete before the call (__syncthreads), `binstart` valid after it
__device__ __forceinline__ void wide_bins_scan(const uint32_t *cnt, uint32_t *binstart, unsigned nbins, uint32_t *wave_tot) {
    const unsigned t = threadIdx.x;
    uint32_t v = 0, x = 0;
    if (t < 256) {
        v = t < nbins ? cnt[t] : 0u;
        x = v;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(x, off, 64);
            if ((int)(t & 63u) >= off) x += y;
        }
        if ((t & 63u) == 63u) wave_tot[t >> 6] = x;
    }
    __syncthreads();
    if (t < nbins) {
        uint32_t base = 0;
        for (unsigned w = 0; w < (t >> 6); w++) base += wave_tot[w];
        binstart[t] = base + x - v;
    }
    __syncthreads();
}

// ---- level 0 + level-1 histogram: digits of every window of a chunk of terms -------------------------------------------------
// digits[w * n + i] = 0 (digit 0: no entry), or (bucket + 1) | sign << 31 with bucket = |digit| - 1 in [0, 2^(c-1)).
// hist1[cidx * nbins1 + bin]: bin = (w << hb1) + h, cidx = chunk  (one bucket set per window), or bin = h, cidx = w * nchunks + chunk.
__global__ void __launch_bounds__(kWideBlock) msmw_digits_hist_kernel(const void *__restrict__ scalars, size_t n, WidePlan p,
                                                                      uint32_t *__restrict__ digits, uint32_t *__restrict__ hist1) {
    extern __shared__ uint32_t cnt[];                   // nwin << hb1
    const unsigned c = p.c, nwin = p.nwin, nbl = nwin << p.hb1;
    for (unsigned b = threadIdx.x; b < nbl; b += blockDim.x) cnt[b] = 0;
    __syncthreads();
    const unsigned chunk = blockIdx.x;
    const size_t lo = (size_t)chunk * p.chunk_len1, hi = lo + p.chunk_len1 < n ? lo + p.chunk_len1 : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        Fe<Fr381> k = fe_to_canonical<Fr381>(fe_load<Fr381>(scalars, i));     // into_bigint()  multilinear_kzg.rs:41
        unsigned carry = 0;
        for (unsigned w = 0; w < nwin; w++) {
            const unsigned bit = w * c, limb = bit >> 5, sh = bit & 31;
            const uint64_t two = (limb < 8 ? (uint64_t)k.l[limb] : 0) | ((uint64_t)(limb + 1 < 8 ? k.l[limb + 1] : 0) << 32);
            const unsigned d = (unsigned)((two >> sh) & ((1u << c) - 1u)) + carry;
            unsigned mag, neg;
            if (d >= (1u << (c - 1)) && w + 1 < nwin) { mag = (1u << c) - d; carry = 1; neg = 1; }   // d - 2^c, in [-2^(c-1), 0]
            else { mag = d; carry = 0; neg = 0; }                                                 // the top window never goes negative
            const uint32_t enc = mag ? (mag | (neg << 31)) : 0u;                                  // mag = bucket + 1
            digits[(size_t)w * n + i] = enc;
            if (mag) atomicAdd(&cnt[(w << p.hb1) + ((mag - 1u) >> p.kb1)], 1u);
        }
    }
    __syncthreads();
    for (unsigned b = threadIdx.x; b < nbl; b += blockDim.x) {
        const unsigned w = b >> p.hb1, h = b & ((1u << p.hb1) - 1u);
        const size_t slot = p.one_set ? ((size_t)w * p.nchunks1 + chunk) * p.nbins1 + h : (size_t)chunk * p.nbins1 + b;
        hist1[slot] = cnt[b];
    }
}

// ---- level-1 scan: absolute start of every (chunk, bin) run; pstart1[bin] = start of the bin's part -----------------------------
// hist1 is chunk-major (row = chunk, column = bin).  Three steps: column sums of row slices (grid), scan over slices and bins (one
// workgroup, <= 2048 bins x <= 64 slices), offsets (grid).  r3: as ONE workgroup walking whole columns it took 0.25 ms with 1664
// bins and 1.0 ms with the 128 bins x 3072 rows of the one-bucket-set layout.
constexpr unsigned kWideScanRows = 64;        // rows per slice
__global__ void __launch_bounds__(256) msmw_scan1_sums_kernel(const uint32_t *__restrict__ hist1, unsigned nbins, unsigned ctot,
                                                              uint64_t *__restrict__ slice_tot) {
    const unsigned b = blockIdx.x * blockDim.x + threadIdx.x, sl = blockIdx.y;
    if (b >= nbins) return;
    const unsigned r0 = sl * kWideScanRows, r1 = r0 + kWideScanRows < ctot ? r0 + kWideScanRows : ctot;
    uint64_t s = 0;
    for (unsigned k = r0; k < r1; k++) s += hist1[(size_t)k * nbins + b];
    slice_tot[(size_t)sl * nbins + b] = s;
}
__global__ void __launch_bounds__(kWideBlock) msmw_scan1_bins_kernel(uint64_t *__restrict__ slice_tot, unsigned nbins, unsigned nslices,
                                                                     uint64_t *__restrict__ pstart1) {
    __shared__ uint64_t tot[2048 + 1];
    for (unsigned b = threadIdx.x; b < nbins; b += blockDim.x) {           // column b: exclusive scan over its slices, total
        uint64_t run = 0;
        for (unsigned sl = 0; sl < nslices; sl++) { const uint64_t v = slice_tot[(size_t)sl * nbins + b]; slice_tot[(size_t)sl * nbins + b] = run; run += v; }
        tot[b] = run;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                   // <= 2048 bins: a short serial prefix, once per MSM
        uint64_t run = 0;
        for (unsigned b = 0; b < nbins; b++) { const uint64_t v = tot[b]; tot[b] = run; run += v; }
        tot[nbins] = run;
    }
    __syncthreads();
    for (unsigned b = threadIdx.x; b <= nbins; b += blockDim.x) pstart1[b] = tot[b];
}
__global__ void __launch_bounds__(256) msmw_scan1_offsets_kernel(const uint32_t *__restrict__ hist1, unsigned nbins, unsigned ctot,
                                                                 const uint64_t *__restrict__ slice_tot, const uint64_t *__restrict__ pstart1,
                                                                 uint64_t *__restrict__ off1) {
    const unsigned b = blockIdx.x * blockDim.x + threadIdx.x, sl = blockIdx.y;
    if (b >= nbins) return;
    const unsigned r0 = sl * kWideScanRows, r1 = r0 + kWideScanRows < ctot ? r0 + kWideScanRows : ctot;
    uint64_t run = pstart1[b] + slice_tot[(size_t)sl * nbins + b];
    for (unsigned k = r0; k < r1; k++) { off1[(size_t)k * nbins + b] = run; run += hist1[(size_t)k * nbins + b]; }
}

// ---- level-1 scatter: workgroup = (chunk, window); tile-sorted whole-run writes ------------------------------------------------------
__global__ void __launch_bounds__(kWideBlock) msmw_l1_scatter_kernel(const uint32_t *__restrict__ digits, size_t n, WidePlan p,
                                                                     const uint64_t *__restrict__ off1, uint32_t *__restrict__ e1,
                                                                     uint16_t *__restrict__ k1) {
    __shared__ uint64_t cursor[128];
    __shared__ uint32_t cnt[128], binstart[128], wave_tot[4];
    __shared__ uint32_t stage_e[kWideTile];
    __shared__ uint16_t stage_k[kWideTile];
    __shared__ uint8_t stage_h[kWideTile];
    constexpr int PER = kWideTile / kWideBlock;
    const unsigned nh = 1u << p.hb1, chunk = blockIdx.x % p.nchunks1, w = blockIdx.x / p.nchunks1;
    const unsigned cidx = p.one_set ? w * p.nchunks1 + chunk : chunk;
    for (unsigned h = threadIdx.x; h < nh; h += blockDim.x)
        cursor[h] = off1[(size_t)cidx * p.nbins1 + (p.one_set ? h : (w << p.hb1) + h)];
    const size_t lo = (size_t)chunk * p.chunk_len1, hi = lo + p.chunk_len1 < n ? lo + p.chunk_len1 : n;
    const uint32_t *d = digits + (size_t)w * n;
    const uint32_t ebase = p.one_set ? (uint32_t)((size_t)w * n) : 0u;
    const uint32_t kmask = (1u << p.kb1) - 1u;
    for (size_t base = lo; base < hi; base += kWideTile) {
        for (unsigned h = threadIdx.x; h < nh; h += blockDim.x) cnt[h] = 0;
        __syncthreads();
        uint32_t ent[PER], rank[PER];
        uint16_t key[PER];
        uint8_t bin[PER];                                  // 0xff = no entry
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const size_t i = base + (size_t)k * kWideBlock + threadIdx.x;
            bin[k] = 0xffu;
            if (i < hi) {
                const uint32_t enc = d[i];
                if (enc) {
                    const uint32_t b = (enc & 0x7fffffffu) - 1u;
                    ent[k] = (ebase + (uint32_t)i) | (enc & 0x80000000u);
                    key[k] = (uint16_t)(b & kmask);
                    bin[k] = (uint8_t)(b >> p.kb1);
                    rank[k] = atomicAdd(&cnt[bin[k]], 1u);
                }
            }
        }
        __syncthreads();
        wide_bins_scan(cnt, binstart, nh, wave_tot);
#pragma unroll
        for (int k = 0; k < PER; k++)
            if (bin[k] != 0xffu) {
                const unsigned pos = binstart[bin[k]] + rank[k];
                stage_e[pos] = ent[k];
                stage_k[pos] = key[k];
                stage_h[pos] = bin[k];
            }
        __syncthreads();
        const uint32_t total = binstart[nh - 1] + cnt[nh - 1];
        for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
            const unsigned h = stage_h[i];
            const uint64_t dst = cursor[h] + (i - binstart[h]);
            e1[dst] = stage_e[i];
            k1[dst] = stage_k[i];
        }
        __syncthreads();
        for (unsigned h = threadIdx.x; h < nh; h += blockDim.x) cursor[h] += cnt[h];
        __syncthreads();
    }
}

// the slice of part [s, e) that workgroup j of c2 owns
__device__ __forceinline__ void wide_slice(uint64_t s, uint64_t e, unsigned j, unsigned c2, uint64_t &lo, uint64_t &hi) {
    const uint64_t per = (e - s + c2 - 1) / c2;
    lo = s + (uint64_t)j * per;
    if (lo > e) lo = e;
    hi = lo + per < e ? lo + per : e;
}

// ---- level 2: histogram of the middle bits, workgroup = (part, slice) ----------------------------------------------------------------
__global__ void __launch_bounds__(256) msmw_l2_hist_kernel(const uint16_t *__restrict__ k1, const uint64_t *__restrict__ pstart1, WidePlan p,
                                                           uint32_t *__restrict__ hist2) {
    __shared__ uint32_t cnt[256];
    const unsigned part = blockIdx.x / p.c2, j = blockIdx.x % p.c2;
    cnt[threadIdx.x] = 0;
    __syncthreads();
    uint64_t lo, hi;
    wide_slice(pstart1[part], pstart1[part + 1], j, p.c2, lo, hi);
    // eight loads in flight per lane: one key per iteration is a chain of load latencies (r3: 0.67 ms for 2 x 10^8 two-byte keys)
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 8 * (uint64_t)blockDim.x) {
        uint16_t k[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint64_t j = i + (uint64_t)u * blockDim.x; k[u] = j < hi ? k1[j] : (uint16_t)0xffffu; }
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint64_t j = i + (uint64_t)u * blockDim.x; if (j < hi) atomicAdd(&cnt[k[u] >> p.lb3], 1u); }
    }
    __syncthreads();
    if (threadIdx.x < p.nb2) hist2[((size_t)part * p.nb2 + threadIdx.x) * p.c2 + j] = cnt[threadIdx.x];
}
// level-2 scan: workgroup per part (256 threads = bins): starts of every (bin, slice) run inside the part, and of every group
__global__ void __launch_bounds__(256) msmw_scan2_kernel(const uint32_t *__restrict__ hist2, const uint64_t *__restrict__ pstart1, WidePlan p,
                                                         uint64_t *__restrict__ off2, uint64_t *__restrict__ pstart2) {
    __shared__ uint32_t cnt[256], binstart[256], wave_tot[4];
    const unsigned part = blockIdx.x, t = threadIdx.x;
    uint32_t s = 0;
    if (t < p.nb2) {
        const uint32_t *h = hist2 + ((size_t)part * p.nb2 + t) * p.c2;
        for (unsigned k = 0; k < p.c2; k++) s += h[k];
    }
    cnt[t] = s;
    __syncthreads();
    wide_bins_scan(cnt, binstart, p.nb2, wave_tot);
    if (t < p.nb2) {
        uint64_t run = pstart1[part] + binstart[t];
        pstart2[(size_t)part * p.nb2 + t] = run;
        const uint32_t *h = hist2 + ((size_t)part * p.nb2 + t) * p.c2;
        uint64_t *o = off2 + ((size_t)part * p.nb2 + t) * p.c2;
        for (unsigned k = 0; k < p.c2; k++) { o[k] = run; run += h[k]; }
    }
    if (part + 1 == gridDim.x && t == 0) pstart2[(size_t)gridDim.x * p.nb2] = pstart1[gridDim.x];
}
// level-2 scatter: workgroup = (part, slice), tile-sorted by the middle bits
__global__ void __launch_bounds__(kWideBlock) msmw_l2_scatter_kernel(const uint32_t *__restrict__ e1, const uint16_t *__restrict__ k1,
                                                                     const uint64_t *__restrict__ pstart1, WidePlan p,
                                                                     const uint64_t *__restrict__ off2, uint32_t *__restrict__ e2,
                                                                     uint8_t *__restrict__ k2) {
    __shared__ uint64_t cursor[256];
    __shared__ uint32_t cnt[256], binstart[256], wave_tot[4];
    __shared__ uint32_t stage_e[kWideTile];
    __shared__ uint8_t stage_k[kWideTile], stage_h[kWideTile];
    constexpr int PER = kWideTile / kWideBlock;
    const unsigned part = blockIdx.x / p.c2, j = blockIdx.x % p.c2, nb2 = p.nb2;
    for (unsigned h = threadIdx.x; h < nb2; h += blockDim.x) cursor[h] = off2[((size_t)part * nb2 + h) * p.c2 + j];
    uint64_t lo, hi;
    wide_slice(pstart1[part], pstart1[part + 1], j, p.c2, lo, hi);
    const unsigned lmask = (1u << p.lb3) - 1u;
    for (uint64_t base = lo; base < hi; base += kWideTile) {
        for (unsigned h = threadIdx.x; h < nb2; h += blockDim.x) cnt[h] = 0;
        __syncthreads();
        uint32_t ent[PER], rank[PER];
        uint16_t key[PER];
        bool have[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const uint64_t i = base + (uint64_t)k * kWideBlock + threadIdx.x;
            have[k] = i < hi;
            if (have[k]) {
                key[k] = k1[i];
                ent[k] = e1[i];
                rank[k] = atomicAdd(&cnt[key[k] >> p.lb3], 1u);
            }
        }
        __syncthreads();
        wide_bins_scan(cnt, binstart, nb2, wave_tot);
#pragma unroll
        for (int k = 0; k < PER; k++)
            if (have[k]) {
                const unsigned h = key[k] >> p.lb3, pos = binstart[h] + rank[k];
                stage_e[pos] = ent[k];
                stage_k[pos] = (uint8_t)(key[k] & lmask);
                stage_h[pos] = (uint8_t)h;
            }
        __syncthreads();
        const uint32_t total = binstart[nb2 - 1] + cnt[nb2 - 1];
        for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
            const unsigned h = stage_h[i];
            const uint64_t dst = cursor[h] + (i - binstart[h]);
            e2[dst] = stage_e[i];
            k2[dst] = stage_k[i];
        }
        __syncthreads();
        for (unsigned h = threadIdx.x; h < nb2; h += blockDim.x) cursor[h] += cnt[h];
        __syncthreads();
    }
}

// ---- level 3: one workgroup per group of 2^lb3 buckets: final order + per-bucket counts -----------------------------------------------
// KT = uint8_t (after level 2) or uint16_t (no level 2: the group is a level-1 part and the key has <= 8 significant bits).
constexpr size_t kWideFineLds = (size_t)kWideFineTile * 5 + (3 * 256 + 4) * 4;
template <class KT>
__global__ void __launch_bounds__(kWideBlock) msmw_l3_kernel(const uint32_t *__restrict__ ein, const KT *__restrict__ kin,
                                                             const uint64_t *__restrict__ gstart, unsigned lb3,
                                                             uint32_t *__restrict__ sorted, uint32_t *__restrict__ totals) {
    extern __shared__ uint32_t lds[];
    uint32_t *stage_e = lds;                                  // [kWideFineTile]
    uint32_t *cnt = lds + kWideFineTile, *binstart = cnt + 256, *cursor = binstart + 256, *wave_tot = cursor + 256;
    uint8_t *stage_b = reinterpret_cast<uint8_t *>(wave_tot + 4);   // [kWideFineTile]
    constexpr int PER = kWideFineTile / kWideBlock;
    const unsigned nl = 1u << lb3;
    const size_t g = blockIdx.x;
    const uint64_t s = gstart[g], e = gstart[g + 1];
    if (e == s) {                                             // an empty group still owns its counts
        for (unsigned l = threadIdx.x; l < nl; l += blockDim.x) totals[g * nl + l] = 0;
        return;
    }
    const bool one_tile = e - s <= (uint64_t)kWideFineTile;
    if (!one_tile) {                                        // heavy group (skewed scalars): count first, then tile by tile
        for (unsigned l = threadIdx.x; l < 256; l += blockDim.x) cnt[l] = 0;
        __syncthreads();
        for (uint64_t i = s + threadIdx.x; i < e; i += 8 * (uint64_t)blockDim.x) {      // eight loads in flight per lane
            KT k[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint64_t j = i + (uint64_t)u * blockDim.x; k[u] = j < e ? kin[j] : (KT)0; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint64_t j = i + (uint64_t)u * blockDim.x; if (j < e) atomicAdd(&cnt[k[u]], 1u); }
        }
        __syncthreads();
        wide_bins_scan(cnt, cursor, nl, wave_tot);            // cursor[l] = offset of bucket l inside the group
        for (unsigned l = threadIdx.x; l < nl; l += blockDim.x) totals[g * nl + l] = cnt[l];
        __syncthreads();
    } else {
        for (unsigned l = threadIdx.x; l < 256; l += blockDim.x) cursor[l] = 0;
    }
    for (uint64_t base = s; base < e; base += kWideFineTile) {
        for (unsigned l = threadIdx.x; l < 256; l += blockDim.x) cnt[l] = 0;
        __syncthreads();
        uint32_t ent[PER], rank[PER];
        uint16_t low[PER];                                   // 0xffff = past the end
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const uint64_t i = base + (uint64_t)k * kWideBlock + threadIdx.x;
            low[k] = 0xffffu;
            if (i < e) {
                low[k] = (uint16_t)kin[i];
                ent[k] = ein[i];
                rank[k] = atomicAdd(&cnt[low[k]], 1u);
            }
        }
        __syncthreads();
        wide_bins_scan(cnt, binstart, nl, wave_tot);
        if (one_tile)                                         // the tile's counts are the group's
            for (unsigned l = threadIdx.x; l < nl; l += blockDim.x) totals[g * nl + l] = cnt[l];
#pragma unroll
        for (int k = 0; k < PER; k++)
            if (low[k] != 0xffffu) {
                const unsigned pos = binstart[low[k]] + rank[k];
                stage_e[pos] = ent[k];
                stage_b[pos] = (uint8_t)low[k];
            }
        __syncthreads();
        const uint32_t total = binstart[nl - 1] + cnt[nl - 1];
        if (one_tile) {                                       // already in final order: one coalesced copy
            for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) sorted[s + i] = stage_e[i];
        } else {
            for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
                const unsigned l = stage_b[i];
                sorted[s + cursor[l] + (i - binstart[l])] = stage_e[i];
            }
            __syncthreads();
            for (unsigned l = threadIdx.x; l < nl; l += blockDim.x) cursor[l] += cnt[l];
        }
        __syncthreads();
    }
}

}  // namespace zk
