// msm_kernels.cuh -- Pippenger multi-scalar multiplication over BLS12-381 G1 for gfx950.
//
// The reference has NO MSM: commit_to_polynomial is a naive sum of mul_bigint terms
// (multilinear_kzg/src/multilinear_kzg.rs:37-42, ~380 group ops per term).  This is the build's
// algorithm for the same group element  C = sum_i [s_i] B_i :
//   1. digits        s_i (Montgomery) -> canonical 255-bit integer -> W signed c-bit digits
//   2. counting sort of (digit, i) per window; the 2^(c-1) bucket counters of one window are
//      staged in LDS (<= 128 KiB of the CU's 160 KiB), so there are no global atomics:
//      histogram per (chunk, window) -> column scan -> scatter with LDS cursors
//   3. bucket sums   one lane per bucket segment: XYZZ accumulator in VGPRs, mixed adds of
//      gathered affine bases (the only HBM-heavy step: 96 B random gather per add)
//   4. bucket reduce S_w = sum_b b * Bucket[w][b] by log-depth halving, in place:
//          A'[b] = A[b] + A[b+H],   R'[b] = A[b+H] + 2 (R[b] + R[b+H])      (H = half)
//      invariant  S_w = f(A) + len * plain(R),  f(A) = sum_b b A[b];  after c levels  S_w = R[0]
//   5. the W window sums are combined on the host (Horner, c doublings per window).
// G1-adds per term = W = ceil(256 / c) mixed adds; integer-VALU bound, not HBM bound.
#pragma once
#include "g1.cuh"
#include "mle_kernels.cuh"

namespace zk {

constexpr int kSortBlock = 1024;   // one workgroup per CU: its LDS holds a whole window's counters

// digit encoding (u16): 0 = skip (digit 0); otherwise bit 15 = sign, low 15 bits = |d| mod 2^15
// (|d| = 2^(c-1) only occurs with a negative sign and is stored as low bits 0).
__device__ __forceinline__ unsigned digit_bucket(unsigned enc, unsigned c) {
    unsigned mag = enc & 0x7fffu;
    return mag ? mag : (1u << (c - 1));
}

#ifdef ZK_MSM_LIGHT_KERNELS   // defined by the one translation unit that launches them (zkmle_kzg.hip)
// scalars: n Fr elements (Montgomery).  digits[w * n + i]
__global__ void msm_digits_kernel(const void *__restrict__ scalars, size_t n, unsigned c, unsigned nwin,
                                  uint16_t *__restrict__ digits) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Fe<Fr381> k = fe_to_canonical<Fr381>(fe_load<Fr381>(scalars, i));     // into_bigint()  multilinear_kzg.rs:41
        unsigned carry = 0;
        for (unsigned w = 0; w < nwin; w++) {
            unsigned bit = w * c;
            unsigned limb = bit >> 5, sh = bit & 31;
            uint64_t two = (limb < 8 ? (uint64_t)k.l[limb] : 0) | ((uint64_t)(limb + 1 < 8 ? k.l[limb + 1] : 0) << 32);
            unsigned d = (unsigned)((two >> sh) & ((1u << c) - 1u)) + carry;
            unsigned enc;
            if (d >= (1u << (c - 1)) && w + 1 < nwin) {             // take d - 2^c (negative) and carry
                unsigned mag = (1u << c) - d;                       // in [0, 2^(c-1)]; 0 when d = 2^c
                carry = 1;
                enc = mag == 0 ? 0u : (0x8000u | (mag & 0x7fffu));
            } else {                                                // positive digit; the top window never goes negative
                carry = 0;                                          // (s < 2^255 keeps it <= 2^(c-1))
                enc = d == 0 ? 0u : ((d & 0x7fffu) | (d == 0x8000u ? 0x8000u : 0u));
            }
            digits[(size_t)w * n + i] = (uint16_t)enc;
        }
    }
}

// pass 1: per (chunk, window) histogram in LDS -> hist[(w * nchunks + chunk) * nb + b]
// `by_chunk`: the output slot of block (w, chunk) is (chunk, w) instead of (w, chunk): every chunk is a bucket set of its
// own that ALL windows feed (batched MSMs on window-shifted bases, zkmle_kzg.hip msm_core)
// `by_chunk` = g > 0: the digit stream is the g halving levels of a KZG opening laid end to end (level j = 2^(g-1-j) entries
// from 2^g - 2^(g-j)), every level is a bucket set of its own, and the chunks are: 2^kSubBits-entry pieces of the levels that
// have at least that many entries (so the big levels spread over many workgroups), then one chunk per smaller level.
constexpr unsigned kSubBits = 15;   // measured r1 (2^20 opening, scatter + scan, ms): 13 -> 1.07, 15 -> 0.38, 16 -> 0.45; one chunk per level 0.93
__device__ __host__ inline unsigned msm_geo_nbig(unsigned g) { return g > kSubBits ? (1u << (g - kSubBits)) - 1u : 0u; }
__device__ __host__ inline unsigned msm_geo_nchunks(unsigned g) { return msm_geo_nbig(g) + (g < kSubBits ? g : kSubBits); }
__device__ __forceinline__ void msm_chunk_range(unsigned chunk, size_t chunk_len, size_t n, unsigned by_chunk, size_t &lo, size_t &hi, unsigned &set) {
    if (!by_chunk) {
        lo = (size_t)chunk * chunk_len;
        hi = lo + chunk_len < n ? lo + chunk_len : n;
        set = 0;
        return;
    }
    const unsigned g = by_chunk, nbig = msm_geo_nbig(g);
    if (chunk < nbig) {
        lo = (size_t)chunk << kSubBits;
        hi = lo + ((size_t)1 << kSubBits);
        set = g - (64u - (unsigned)__clzll((long long)((((size_t)1 << g) - lo) - 1)));      // g - ceil(log2(2^g - lo))
    } else {
        set = (g > kSubBits ? g - kSubBits : 0u) + (chunk - nbig);
        lo = ((size_t)1 << g) - ((size_t)1 << (g - set));
        hi = lo + ((size_t)1 << (g - 1 - set));
    }
}
__global__ void msm_hist_kernel(const uint16_t *__restrict__ digits, size_t n, unsigned c, unsigned nchunks,
                                size_t chunk_len, uint32_t *__restrict__ hist, unsigned by_chunk) {
    extern __shared__ uint32_t lds[];
    unsigned nb = 1u << (c - 1);
    unsigned chunk = blockIdx.x % nchunks, w = blockIdx.x / nchunks;
    for (unsigned b = threadIdx.x; b < nb; b += blockDim.x) lds[b] = 0;
    __syncthreads();
    size_t lo, hi;
    unsigned set;
    msm_chunk_range(chunk, chunk_len, n, by_chunk, lo, hi, set);
    const uint16_t *d = digits + (size_t)w * n;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        unsigned enc = d[i];
        if (enc) atomicAdd(&lds[digit_bucket(enc, c) - 1], 1u);
    }
    __syncthreads();
    const unsigned nwin = gridDim.x / nchunks;
    uint32_t *out = hist + (by_chunk ? (size_t)chunk * nwin + w : (size_t)w * nchunks + chunk) * nb;
    for (unsigned b = threadIdx.x; b < nb; b += blockDim.x) out[b] = lds[b];
}

// pass 2a: per (w, b): exclusive prefix over chunks (in place) and the bucket total
__global__ void msm_chunk_scan_kernel(uint32_t *__restrict__ hist, unsigned nwin, unsigned nchunks, unsigned nb,
                                      uint32_t *__restrict__ totals) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)nwin * nb) return;
    unsigned w = id / nb, b = id % nb;
    uint32_t run = 0;
    for (unsigned ch = 0; ch < nchunks; ch++) {
        uint32_t *p = hist + ((size_t)w * nchunks + ch) * nb + b;
        uint32_t v = *p;
        *p = run;
        run += v;
    }
    totals[id] = run;
}

// pass 2a for the level layout (by_chunk = g): bucket set j is fed by the rows (chunk, w) of level j's chunks, which are
// contiguous: exclusive prefix over them (in place) and the bucket total
__global__ void msm_set_scan_kernel(uint32_t *__restrict__ hist, unsigned g, unsigned nwin1, unsigned nb, uint32_t *__restrict__ totals) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)g * nb) return;
    unsigned j = id / nb, b = id % nb;
    const unsigned nbig = msm_geo_nbig(g), jb = g > kSubBits ? g - kSubBits : 0u;
    size_t chunk_lo, cnt;
    if (j < jb) { chunk_lo = (((size_t)1 << g) - ((size_t)1 << (g - j))) >> kSubBits; cnt = (size_t)1 << (g - 1 - j - kSubBits); }
    else { chunk_lo = nbig + (j - jb); cnt = 1; }
    uint32_t run = 0;
    for (size_t r = chunk_lo * nwin1; r < (chunk_lo + cnt) * nwin1; r++) {
        uint32_t *p = hist + r * nb + b;
        uint32_t v = *p;
        *p = run;
        run += v;
    }
    totals[id] = run;
}

// pass 2b: exclusive scans over all (w, b) of the entry counts and of the segment counts
// ceil(count / seg_len), as three coalesced kernels: per-tile totals, scan of the tile totals (one block),
// per-tile scan + offset.  starts[count] / seg_starts[count] hold the grand totals and
// seg_starts[count + 1] the largest per-bucket segment count.
constexpr int kScanTile = 1024;
__global__ void msm_scan_tiles_kernel(const uint32_t *__restrict__ totals, size_t count, unsigned seg_len,
                                      uint64_t *__restrict__ tile_e, uint32_t *__restrict__ tile_s, uint32_t *__restrict__ tile_m) {
    __shared__ uint64_t sh_e[kScanTile / 64];
    __shared__ uint32_t sh_s[kScanTile / 64], sh_m[kScanTile / 64];
    size_t i = (size_t)blockIdx.x * kScanTile + threadIdx.x;
    uint32_t v = i < count ? totals[i] : 0u;
    uint32_t segs = (v + seg_len - 1) / seg_len;
    uint64_t e = v;
    uint32_t s = segs, m = segs;
    for (int off = 32; off >= 1; off >>= 1) {
        e += __shfl_down(e, off, 64);
        s += __shfl_down(s, off, 64);
        uint32_t om = __shfl_down(m, off, 64);
        m = om > m ? om : m;
    }
    if ((threadIdx.x & 63) == 0) { sh_e[threadIdx.x >> 6] = e; sh_s[threadIdx.x >> 6] = s; sh_m[threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t te = 0; uint32_t ts = 0, tm = 0;
        for (int w = 0; w < kScanTile / 64; w++) { te += sh_e[w]; ts += sh_s[w]; tm = sh_m[w] > tm ? sh_m[w] : tm; }
        tile_e[blockIdx.x] = te; tile_s[blockIdx.x] = ts; tile_m[blockIdx.x] = tm;
    }
}
// exclusive scan of the per-tile totals (entries, segments) and the largest per-bucket segment count: ONE wave, lane l owns a contiguous
// run of tiles (r1: one lane walked all of them, a chain of dependent global loads: 57 us for the 2048 tiles of a 2^20-term MSM)
__global__ void msm_scan_tile_totals_kernel(uint64_t *__restrict__ tile_e, uint32_t *__restrict__ tile_s, const uint32_t *__restrict__ tile_m,
                                            unsigned ntiles, size_t count, uint64_t *__restrict__ starts, uint32_t *__restrict__ seg_starts) {
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    const unsigned lane = threadIdx.x, per = (ntiles + 63u) / 64u;
    const unsigned lo = lane * per < ntiles ? lane * per : ntiles, hi = lo + per < ntiles ? lo + per : ntiles;
    uint64_t se = 0; uint32_t ss = 0, rm = 0;
    for (unsigned t = lo; t < hi; t++) { se += tile_e[t]; ss += tile_s[t]; rm = tile_m[t] > rm ? tile_m[t] : rm; }
    uint64_t pe = se; uint32_t ps = ss;                       // inclusive scan over the lanes
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t ue = __shfl_up(pe, off);
        const uint32_t us = __shfl_up(ps, off);
        const uint32_t um = __shfl_xor(rm, off);
        if ((int)lane >= off) { pe += ue; ps += us; }
        rm = um > rm ? um : rm;                               // butterfly: every lane ends with the maximum
    }
    uint64_t re = pe - se; uint32_t rs = ps - ss;             // exclusive prefix of this lane's run
    for (unsigned t = lo; t < hi; t++) {
        const uint64_t ve = tile_e[t]; const uint32_t vs = tile_s[t];
        tile_e[t] = re; tile_s[t] = rs;
        re += ve; rs += vs;
    }
    if (lane == 63) {
        starts[count] = pe;
        seg_starts[count] = ps;
        seg_starts[count + 1] = rm;
    }
}
__global__ void msm_scan_apply_kernel(const uint32_t *__restrict__ totals, size_t count, unsigned seg_len,
                                      const uint64_t *__restrict__ tile_e, const uint32_t *__restrict__ tile_s,
                                      uint64_t *__restrict__ starts, uint32_t *__restrict__ seg_starts) {
    __shared__ uint64_t sh_e[kScanTile];
    __shared__ uint32_t sh_s[kScanTile];
    size_t i = (size_t)blockIdx.x * kScanTile + threadIdx.x;
    uint32_t v = i < count ? totals[i] : 0u;
    sh_e[threadIdx.x] = v;
    sh_s[threadIdx.x] = (v + seg_len - 1) / seg_len;
    __syncthreads();
    for (int off = 1; off < kScanTile; off <<= 1) {      // Hillis-Steele inclusive scan of the tile
        uint64_t ae = threadIdx.x >= (unsigned)off ? sh_e[threadIdx.x - off] : 0;
        uint32_t as = threadIdx.x >= (unsigned)off ? sh_s[threadIdx.x - off] : 0;
        __syncthreads();
        sh_e[threadIdx.x] += ae;
        sh_s[threadIdx.x] += as;
        __syncthreads();
    }
    if (i < count) {
        starts[i] = tile_e[blockIdx.x] + sh_e[threadIdx.x] - v;
        seg_starts[i] = tile_s[blockIdx.x] + sh_s[threadIdx.x] - (v + seg_len - 1) / seg_len;
    }
}

// ---- partial-sum slots of the bucket kernel (msm_bucket.hip): a lane takes a RUN of `run` consecutive sorted entries, so bucket b
// with range [s, s + n) gets one partial per run it overlaps: segs(b) = (s + n - 1) / run - s / run + 1 (0 when empty).  seg_starts =
// exclusive scan of segs over all buckets; seg_starts[count] = total, seg_starts[count + 1] = the largest segs(b).  Same three-step
// shape as the scan of the counts above, run after it (segs needs the bucket starts).
struct SegFromRuns {                                     // partial slots of the bucket kernel
    const uint64_t *starts;
    unsigned run;
    __device__ __forceinline__ uint32_t operator()(size_t i) const {
        const uint64_t s = starts[i], e = starts[i + 1];
        return e > s ? (uint32_t)((e - 1) / run - s / run + 1) : 0u;
    }
};
struct SegFromGroups {                                   // regrouping: sums of up to `group` consecutive partials of a bucket
    const uint32_t *in_starts;
    unsigned group;
    __device__ __forceinline__ uint32_t operator()(size_t i) const { return (in_starts[i + 1] - in_starts[i] + group - 1) / group; }
};
template <class Fn>
__global__ void msm_seg_tiles_kernel(Fn fn, size_t count, uint32_t *__restrict__ tile_s, uint32_t *__restrict__ tile_m) {
    __shared__ uint32_t sh_s[kScanTile / 64], sh_m[kScanTile / 64];
    const size_t i = (size_t)blockIdx.x * kScanTile + threadIdx.x;
    uint32_t s = i < count ? fn(i) : 0u, m = s;
    for (int off = 32; off >= 1; off >>= 1) {
        s += __shfl_down(s, off, 64);
        const uint32_t om = __shfl_down(m, off, 64);
        m = om > m ? om : m;
    }
    if ((threadIdx.x & 63) == 0) { sh_s[threadIdx.x >> 6] = s; sh_m[threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ts = 0, tm = 0;
        for (int w = 0; w < kScanTile / 64; w++) { ts += sh_s[w]; tm = sh_m[w] > tm ? sh_m[w] : tm; }
        tile_s[blockIdx.x] = ts; tile_m[blockIdx.x] = tm;
    }
}
__global__ void msm_seg_totals_kernel(uint32_t *__restrict__ tile_s, const uint32_t *__restrict__ tile_m, unsigned ntiles, size_t count,
                                      uint32_t *__restrict__ seg_starts) {
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    const unsigned lane = threadIdx.x, per = (ntiles + 63u) / 64u;
    const unsigned lo = lane * per < ntiles ? lane * per : ntiles, hi = lo + per < ntiles ? lo + per : ntiles;
    uint32_t ss = 0, rm = 0;
    for (unsigned t = lo; t < hi; t++) { ss += tile_s[t]; rm = tile_m[t] > rm ? tile_m[t] : rm; }
    uint32_t ps = ss;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t us = __shfl_up(ps, off);
        const uint32_t um = __shfl_xor(rm, off);
        if ((int)lane >= off) ps += us;
        rm = um > rm ? um : rm;
    }
    uint32_t rs = ps - ss;
    for (unsigned t = lo; t < hi; t++) { const uint32_t vs = tile_s[t]; tile_s[t] = rs; rs += vs; }
    if (lane == 63) { seg_starts[count] = ps; seg_starts[count + 1] = rm; }
}
template <class Fn>
__global__ void msm_seg_apply_kernel(Fn fn, size_t count, const uint32_t *__restrict__ tile_s, uint32_t *__restrict__ seg_starts) {
    __shared__ uint32_t sh_s[kScanTile];
    const size_t i = (size_t)blockIdx.x * kScanTile + threadIdx.x;
    const uint32_t v = i < count ? fn(i) : 0u;
    sh_s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < kScanTile; off <<= 1) {      // Hillis-Steele inclusive scan of the tile
        const uint32_t as = threadIdx.x >= (unsigned)off ? sh_s[threadIdx.x - off] : 0;
        __syncthreads();
        sh_s[threadIdx.x] += as;
        __syncthreads();
    }
    if (i < count) seg_starts[i] = tile_s[blockIdx.x] + sh_s[threadIdx.x] - v;
}
// seg_starts[0 .. count) = exclusive scan of fn(i); [count] = total, [count + 1] = max.  tile_s / tile_m: ceil(count / kScanTile) words each.
template <class Fn>
inline int msm_seg_scan(Fn fn, size_t count, uint32_t *tile_s, uint32_t *tile_m, uint32_t *seg_starts, hipStream_t s) {
    const unsigned ntiles = (unsigned)((count + kScanTile - 1) / kScanTile);
    msm_seg_tiles_kernel<Fn><<<ntiles, kScanTile, 0, s>>>(fn, count, tile_s, tile_m);
    msm_seg_totals_kernel<<<1, 64, 0, s>>>(tile_s, tile_m, ntiles, count, seg_starts);
    msm_seg_apply_kernel<Fn><<<ntiles, kScanTile, 0, s>>>(fn, count, tile_s, seg_starts);
    return hipGetLastError() == hipSuccess ? ZK_OK : ZK_E_HIP;
}

// pass 3: scatter (index | sign << 31) into bucket order, cursors staged in LDS
__global__ void msm_scatter_kernel(const uint16_t *__restrict__ digits, size_t n, unsigned c, unsigned nchunks,
                                   size_t chunk_len, const uint32_t *__restrict__ hist,
                                   const uint64_t *__restrict__ starts, uint32_t *__restrict__ sorted, unsigned by_chunk) {
    extern __shared__ uint32_t lds[];
    unsigned nb = 1u << (c - 1);
    unsigned chunk = blockIdx.x % nchunks, w = blockIdx.x / nchunks;
    const unsigned nwin = gridDim.x / nchunks;
    // by_chunk (see msm_hist_kernel): bucket set = chunk, fed by every window; the entry names the point of window w's
    // shifted copy of the bases, index w * n + i
    const uint32_t *off = hist + (by_chunk ? (size_t)chunk * nwin + w : (size_t)w * nchunks + chunk) * nb;
    for (unsigned b = threadIdx.x; b < nb; b += blockDim.x) lds[b] = off[b];
    __syncthreads();
    size_t lo, hi;
    unsigned set;
    msm_chunk_range(chunk, chunk_len, n, by_chunk, lo, hi, set);
    const uint16_t *d = digits + (size_t)w * n;
    const uint64_t *st = starts + (size_t)(by_chunk ? set : w) * nb;
    const size_t base = by_chunk ? (size_t)w * n : 0;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        unsigned enc = d[i];
        if (enc) {
            unsigned b = digit_bucket(enc, c) - 1;
            uint32_t pos = atomicAdd(&lds[b], 1u);
            unsigned neg = (enc & 0x8000u) && (w + 1 < nwin);                  // the top window is never negative
            sorted[st[b] + pos] = (uint32_t)(base + i) | (neg << 31);
        }
    }
}

// ---- two-level scatter for wide windows (c >= 12) --------------------------------------------------------------
// The single-pass scatter above writes 4-byte entries to 2^(c-1) open runs per window at once: at 2^24 terms it is
// write-amplification bound (6.4 ms).  Here entries are first partitioned by the HIGH bits of the bucket id
// (<= 128 partitions per window; each workgroup sorts a 2048-element tile in LDS and writes whole runs), then each
// partition (a contiguous range of 256 buckets, ~0.5 MB of entries) is finished by one workgroup whose 256 open
// runs stay L2-resident.  A partition occupies the same index range in the intermediate and in the final array.
constexpr unsigned kFineBits = 8;
constexpr int kPartTile = 8192;      // elements per LDS tile
constexpr int kPartBlock = 1024;

// A1: phist[(w * nh + h) * nchunks + chunk] = entries of chunk whose bucket has high part h
__global__ void msm_part_hist_kernel(const uint16_t *__restrict__ digits, size_t n, unsigned c, unsigned nchunks, size_t chunk_len,
                                     uint32_t *__restrict__ phist) {
    __shared__ uint32_t cnt[128];
    unsigned lb = (c - 1) < kFineBits ? (c - 1) : kFineBits, nh = 1u << (c - 1 - lb);
    unsigned chunk = blockIdx.x % nchunks, w = blockIdx.x / nchunks;
    for (unsigned h = threadIdx.x; h < nh; h += blockDim.x) cnt[h] = 0;
    __syncthreads();
    size_t lo = (size_t)chunk * chunk_len, hi = lo + chunk_len < n ? lo + chunk_len : n;
    const uint16_t *d = digits + (size_t)w * n;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        unsigned enc = d[i];
        if (enc) atomicAdd(&cnt[(digit_bucket(enc, c) - 1) >> lb], 1u);
    }
    __syncthreads();
    for (unsigned h = threadIdx.x; h < nh; h += blockDim.x) phist[((size_t)w * nh + h) * nchunks + chunk] = cnt[h];
}
// A2: per (w, h): absolute start of every chunk's run = partition start + exclusive prefix over chunks
__global__ void msm_part_scan_kernel(uint32_t *__restrict__ phist, unsigned nwin, unsigned c, unsigned nchunks,
                                     const uint64_t *__restrict__ starts, uint64_t *__restrict__ poff) {
    unsigned lb = (c - 1) < kFineBits ? (c - 1) : kFineBits, nh = 1u << (c - 1 - lb), nb = 1u << (c - 1);
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)nwin * nh) return;
    unsigned w = id / nh, h = id % nh;
    uint64_t run = starts[(size_t)w * nb + ((size_t)h << lb)];
    for (unsigned ch = 0; ch < nchunks; ch++) {
        uint32_t v = phist[id * nchunks + ch];
        poff[id * nchunks + ch] = run;
        run += v;
    }
}
// A3: partition the chunk's entries by h with whole-run writes (tile-local counting sort in LDS)
__global__ void __launch_bounds__(kPartBlock) msm_part_scatter_kernel(const uint16_t *__restrict__ digits, size_t n, unsigned c, unsigned nchunks,
                                                                      size_t chunk_len, const uint64_t *__restrict__ poff,
                                                                      uint32_t *__restrict__ part_e, uint8_t *__restrict__ part_l) {
    __shared__ uint64_t cursor[128];
    __shared__ uint32_t cnt[128], binstart[128];
    __shared__ uint32_t stage_e[kPartTile];
    __shared__ uint8_t stage_l[kPartTile], stage_h[kPartTile];
    constexpr int PER = kPartTile / kPartBlock;
    unsigned lb = (c - 1) < kFineBits ? (c - 1) : kFineBits, nh = 1u << (c - 1 - lb);
    unsigned nwin = gridDim.x / nchunks;
    unsigned chunk = blockIdx.x % nchunks, w = blockIdx.x / nchunks;
    for (unsigned h = threadIdx.x; h < nh; h += blockDim.x) cursor[h] = poff[((size_t)w * nh + h) * nchunks + chunk];
    size_t lo = (size_t)chunk * chunk_len, hi = lo + chunk_len < n ? lo + chunk_len : n;
    const uint16_t *d = digits + (size_t)w * n;
    for (size_t base = lo; base < hi; base += kPartTile) {
        for (unsigned h = threadIdx.x; h < nh; h += blockDim.x) cnt[h] = 0;
        __syncthreads();
        uint32_t ent[PER], rank[PER];
        uint16_t hl[PER];                       // h << 8 | low ; 0xffff = skip
#pragma unroll
        for (int k = 0; k < PER; k++) {
            size_t i = base + (size_t)k * kPartBlock + threadIdx.x;
            hl[k] = 0xffffu;
            if (i < hi) {
                unsigned enc = d[i];
                if (enc) {
                    unsigned b = digit_bucket(enc, c) - 1, h = b >> lb;
                    unsigned neg = (enc & 0x8000u) && (w + 1 < nwin);
                    ent[k] = (uint32_t)i | (neg << 31);
                    hl[k] = (uint16_t)((h << 8) | (b & ((1u << lb) - 1u)));
                    rank[k] = atomicAdd(&cnt[h], 1u);
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            for (unsigned h = 0; h < nh; h++) { binstart[h] = run; run += cnt[h]; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; k++)
            if (hl[k] != 0xffffu) {
                unsigned h = hl[k] >> 8, pos = binstart[h] + rank[k];
                stage_e[pos] = ent[k];
                stage_l[pos] = (uint8_t)(hl[k] & 0xffu);
                stage_h[pos] = (uint8_t)h;
            }
        __syncthreads();
        uint32_t total = binstart[nh - 1] + cnt[nh - 1];
        for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
            unsigned h = stage_h[i];
            uint64_t dst = cursor[h] + (i - binstart[h]);
            part_e[dst] = stage_e[i];
            part_l[dst] = stage_l[i];
        }
        __syncthreads();
        for (unsigned h = threadIdx.x; h < nh; h += blockDim.x) cursor[h] += cnt[h];
        __syncthreads();
    }
}
// B: one workgroup per partition (w, h): final placement by the low bits.  Tiles of kFineTile entries are counting-sorted
// in LDS first, so that every bucket's share of a tile (64 entries on average) leaves as one contiguous run; placing entries
// one by one (4-byte stores scattered over the partition) took 2.8 ms at 2^24, L2-transaction bound.
constexpr int kFineTile = 16384;
constexpr size_t kFineLdsBytes = (size_t)kFineTile * 5 + 3 * 256 * 4;
__global__ void __launch_bounds__(kSortBlock) msm_fine_scatter_kernel(const uint32_t *__restrict__ part_e, const uint8_t *__restrict__ part_l, unsigned c,
                                                                      const uint64_t *__restrict__ starts, uint32_t *__restrict__ sorted) {
    extern __shared__ uint32_t lds[];
    uint32_t *stage_e = lds;                                  // [kFineTile]
    uint32_t *cnt = lds + kFineTile, *binstart = cnt + 256, *cursor = binstart + 256;
    uint8_t *stage_b = reinterpret_cast<uint8_t *>(cursor + 256);   // [kFineTile]
    constexpr int PER = kFineTile / kSortBlock;
    unsigned lb = (c - 1) < kFineBits ? (c - 1) : kFineBits, nh = 1u << (c - 1 - lb), nb = 1u << (c - 1);
    unsigned w = blockIdx.x / nh, h = blockIdx.x % nh;
    size_t first_bucket = (size_t)w * nb + ((size_t)h << lb);
    uint64_t pstart = starts[first_bucket], pend = starts[first_bucket + (1u << lb)];
    for (unsigned l = threadIdx.x; l < 256; l += blockDim.x) cursor[l] = l < (1u << lb) ? (uint32_t)(starts[first_bucket + l] - pstart) : 0u;
    for (uint64_t base = pstart; base < pend; base += kFineTile) {
        for (unsigned l = threadIdx.x; l < 256; l += blockDim.x) cnt[l] = 0;
        __syncthreads();
        uint32_t ent[PER], rank[PER];
        uint16_t low[PER];                                   // 0xffff = past the end
#pragma unroll
        for (int k = 0; k < PER; k++) {
            uint64_t i = base + (uint64_t)k * kSortBlock + threadIdx.x;
            low[k] = 0xffffu;
            if (i < pend) {
                low[k] = part_l[i];
                ent[k] = part_e[i];
                rank[k] = atomicAdd(&cnt[low[k]], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            for (unsigned l = 0; l < 256; l++) { binstart[l] = run; run += cnt[l]; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; k++)
            if (low[k] != 0xffffu) {
                unsigned pos = binstart[low[k]] + rank[k];
                stage_e[pos] = ent[k];
                stage_b[pos] = (uint8_t)low[k];
            }
        __syncthreads();
        const uint32_t total = binstart[255] + cnt[255];
        for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
            unsigned l = stage_b[i];
            sorted[pstart + cursor[l] + (i - binstart[l])] = stage_e[i];
        }
        __syncthreads();
        for (unsigned l = threadIdx.x; l < 256; l += blockDim.x) cursor[l] += cnt[l];
        __syncthreads();
    }
}

#endif  // ZK_MSM_LIGHT_KERNELS

// ---- heavy kernels live in their own translation units (compiled in parallel); launchers: ----------
constexpr int kNormPer = 16;
int launch_msm_bucket_sum(const void *bases, const uint32_t *sorted, const uint64_t *starts, const uint32_t *seg_starts,
                          size_t nbuckets, unsigned run, uint64_t entries, void *partials, hipStream_t s);
int launch_msm_partials_regroup(const void *in_partials, const uint32_t *in_starts, const uint32_t *out_starts, size_t nbuckets,
                                unsigned group, uint32_t nout, void *out_partials, hipStream_t s);
constexpr size_t kBaseUBytes = 128;   // one pre-converted affine point (g1u.cuh)
constexpr size_t kXyzzUBytes = 256;   // one XYZZ point in the internal form (g1u.cuh): partial sums, reduction arrays
int launch_msm_plain_level(void *A, void *B, unsigned nwin, unsigned cm1, unsigned k, size_t hh, size_t lh, hipStream_t s);
int launch_msm_gather_cd(const void *A, const void *B, unsigned nwin, unsigned cm1, unsigned k, unsigned mbits, void *X, hipStream_t s);
int launch_msm_two_stage_out(const void *X, const void *Y, unsigned nwin, unsigned mbits, void *out, hipStream_t s);
int launch_msm_weighted_bits(void *X, void *Y, unsigned narrays, unsigned mbits, void *S, hipStream_t s);   // the same sums by the bits of the weight (mbits <= 8; r4)
int launch_msm_weighted_tail(void *X, void *Y, unsigned narrays, unsigned mbits, hipStream_t s);   // all weighted levels of the short arrays (the wide first ones grid-wide)
int launch_msm_window_sums(const void *A, const void *R, unsigned nwin, unsigned c, void *out, hipStream_t s);
int launch_g1_bases_to_u(const void *affine, size_t n, void *out_u, hipStream_t s);
int launch_g1_shift(const void *in, int in_is_xyzz, size_t n, unsigned c, void *out_xyzz, hipStream_t s);   // out = 2^c * in (stored affine or XYZZ in)
int launch_msm_bucket_combine(const void *partials, const uint32_t *seg_starts, unsigned nwin, unsigned c, void *A, void *B, hipStream_t s);
int launch_msm_reduce_level(void *A, void *R, unsigned nwin, unsigned c, size_t half, hipStream_t s);
int launch_g1_pair_add(const void *in_affine, size_t half, void *out_xyzz, hipStream_t s);
int launch_g1_pair_add_xyzz(const void *in_xyzz, size_t half, void *out_xyzz, hipStream_t s);
int launch_fixed_base_mul(const void *scalars, size_t n, const void *table16, void *out_xyzz, hipStream_t s);   // table16: pre-converted (g1u.cuh)
int launch_fixed_table16(const void *table8, void *out_xyzz, hipStream_t s);
int launch_batch_to_affine(const void *xyzz, size_t n, void *affine, hipStream_t s);
int launch_synthetic_bases(const G1Affine &g, const G1Affine &dstep, const Fe<Fr381> &a_canon, const Fe<Fr381> &d_canon, size_t n,
                           unsigned per, void *out_xyzz, hipStream_t s);

}  // namespace zk
