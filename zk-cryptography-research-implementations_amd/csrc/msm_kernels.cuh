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
__global__ void msm_hist_kernel(const uint16_t *__restrict__ digits, size_t n, unsigned c, unsigned nchunks,
                                size_t chunk_len, uint32_t *__restrict__ hist) {
    extern __shared__ uint32_t lds[];
    unsigned nb = 1u << (c - 1);
    unsigned chunk = blockIdx.x % nchunks, w = blockIdx.x / nchunks;
    for (unsigned b = threadIdx.x; b < nb; b += blockDim.x) lds[b] = 0;
    __syncthreads();
    size_t lo = (size_t)chunk * chunk_len, hi = lo + chunk_len < n ? lo + chunk_len : n;
    const uint16_t *d = digits + (size_t)w * n;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        unsigned enc = d[i];
        if (enc) atomicAdd(&lds[digit_bucket(enc, c) - 1], 1u);
    }
    __syncthreads();
    uint32_t *out = hist + ((size_t)w * nchunks + chunk) * nb;
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

// pass 2b (one block): exclusive scans over all (w, b) of the entry counts and of the segment
// counts ceil(count / seg_len).  starts[count] / seg_starts[count] hold the grand totals.
__global__ void msm_bucket_scan_kernel(const uint32_t *__restrict__ totals, size_t count, unsigned seg_len,
                                       uint64_t *__restrict__ starts, uint32_t *__restrict__ seg_starts) {
    __shared__ uint64_t sh_e[kSortBlock];
    __shared__ uint32_t sh_s[kSortBlock];
    size_t per = (count + blockDim.x - 1) / blockDim.x;
    size_t lo = (size_t)threadIdx.x * per, hi = lo + per < count ? lo + per : count;
    uint64_t se = 0;
    uint32_t ss = 0;
    for (size_t i = lo; i < hi; i++) { se += totals[i]; ss += (totals[i] + seg_len - 1) / seg_len; }
    sh_e[threadIdx.x] = se;
    sh_s[threadIdx.x] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t re = 0;
        uint32_t rs = 0;
        for (unsigned t = 0; t < blockDim.x; t++) {
            uint64_t ve = sh_e[t]; uint32_t vs = sh_s[t];
            sh_e[t] = re; sh_s[t] = rs;
            re += ve; rs += vs;
        }
        starts[count] = re;
        seg_starts[count] = rs;
    }
    __syncthreads();
    se = sh_e[threadIdx.x];
    ss = sh_s[threadIdx.x];
    for (size_t i = lo; i < hi; i++) {
        starts[i] = se;
        seg_starts[i] = ss;
        se += totals[i];
        ss += (totals[i] + seg_len - 1) / seg_len;
    }
}

// pass 3: scatter (index | sign << 31) into bucket order, cursors staged in LDS
__global__ void msm_scatter_kernel(const uint16_t *__restrict__ digits, size_t n, unsigned c, unsigned nchunks,
                                   size_t chunk_len, const uint32_t *__restrict__ hist,
                                   const uint64_t *__restrict__ starts, uint32_t *__restrict__ sorted) {
    extern __shared__ uint32_t lds[];
    unsigned nb = 1u << (c - 1);
    unsigned chunk = blockIdx.x % nchunks, w = blockIdx.x / nchunks;
    const uint32_t *off = hist + ((size_t)w * nchunks + chunk) * nb;
    for (unsigned b = threadIdx.x; b < nb; b += blockDim.x) lds[b] = off[b];
    __syncthreads();
    size_t lo = (size_t)chunk * chunk_len, hi = lo + chunk_len < n ? lo + chunk_len : n;
    const uint16_t *d = digits + (size_t)w * n;
    const uint64_t *st = starts + (size_t)w * nb;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        unsigned enc = d[i];
        if (enc) {
            unsigned b = digit_bucket(enc, c) - 1;
            uint32_t pos = atomicAdd(&lds[b], 1u);
            unsigned neg = (enc & 0x8000u) && (w + 1 < gridDim.x / nchunks);   // the top window is never negative
            sorted[st[b] + pos] = (uint32_t)i | (neg << 31);
        }
    }
}

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

// bucket (w, b) = sum of its segments' partials, written to slot b + 1 of window w in the
// 2^c-slot reduction array A (slot index = digit magnitude)
__global__ void __launch_bounds__(256) msm_bucket_combine_kernel(const void *__restrict__ partials, const uint32_t *__restrict__ seg_starts,
                                          unsigned nwin, unsigned c, void *__restrict__ A) {
    unsigned nb = 1u << (c - 1);
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)nwin * nb) return;
    unsigned w = id / nb, b = id % nb;
    uint32_t s0 = seg_starts[id], s1 = seg_starts[id + 1];
    G1Xyzz acc = g1_xyzz_inf();
    for (uint32_t s = s0; s < s1; s++) acc = g1_add_ni(acc, g1_load_xyzz(partials, s));
    g1_store_xyzz(A, ((size_t)w << c) + b + 1, acc);
}

// step 4: one halving level, in place.  half = current length / 2
__global__ void __launch_bounds__(256) msm_reduce_level_kernel(void *__restrict__ A, void *__restrict__ R, unsigned nwin, unsigned c, size_t half) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)nwin * half) return;
    size_t w = id / half, b = id % half;
    size_t base = w << c;
    G1Xyzz alo = g1_load_xyzz(A, base + b), ahi = g1_load_xyzz(A, base + b + half);
    G1Xyzz rlo = g1_load_xyzz(R, base + b), rhi = g1_load_xyzz(R, base + b + half);
    g1_store_xyzz(A, base + b, g1_add_ni(alo, ahi));
    g1_store_xyzz(R, base + b, g1_add_ni(ahi, g1_dbl_ni(g1_add_ni(rlo, rhi))));
}

// out[k] = in[k] + in[k + half]  (affine + affine -> XYZZ): the pre-summed opening bases
// B^(t+1)_k = B^(t)_k + B^(t)_{k + half}  
__global__ void __launch_bounds__(256) g1_pair_add_kernel(const void *__restrict__ in_affine, size_t half, void *__restrict__ out_xyzz) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= half) return;
    g1_store_xyzz(out_xyzz, k, g1_madd_ni(g1_from_affine(g1_load_affine(in_affine, k)), g1_load_affine(in_affine, k + half)));
}

// ---- setup-side kernels ----------------------------------------------------------------------------
// eq / Lagrange table  L_idx(tau) = prod_i (bit_i(idx) ? tau_i : 1 - tau_i), variable 0 = MSB
// (compute_lagrange_basis trusted_setup.rs:24-49), built level by level: out has 2 * len entries,
// out[2 j] = in[j] * (1 - tau), out[2 j + 1] = in[j] * tau.
__global__ void eq_expand_kernel(const void *__restrict__ in, void *__restrict__ out, size_t len, Fe<Fr381> tau) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    Fe<Fr381> omt = fe_sub<Fr381>(fe_one<Fr381>(), tau);
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < len; j += stride) {
        Fe<Fr381> v = fe_load<Fr381>(in, j);
        fe_store<Fr381>(out, 2 * j, fe_mul<Fr381>(v, omt));
        fe_store<Fr381>(out, 2 * j + 1, fe_mul<Fr381>(v, tau));
    }
}

// fixed-base scalar multiplication  out[i] = [s_i] G  with a byte-window table of G
// (table[j * 256 + v] = [v * 256^j] G, affine, 32 x 256 entries): 32 mixed adds per point
// (compute_g1_powers_of_tau trusted_setup.rs:51-60 does one 255-bit double-and-add per point).
__global__ void __launch_bounds__(256) fixed_base_mul_kernel(const void *__restrict__ scalars, size_t n, const void *__restrict__ table,
                                                             void *__restrict__ out_xyzz) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<Fr381> k = fe_to_canonical<Fr381>(fe_load<Fr381>(scalars, i));
    G1Xyzz acc = g1_xyzz_inf();
    for (int j = 0; j < 32; j++) {
        unsigned v = (k.l[j >> 2] >> (8 * (j & 3))) & 0xffu;
        if (v) acc = g1_madd_ni(acc, g1_load_affine(table, (size_t)j * 256 + v));
    }
    g1_store_xyzz(out_xyzz, i, acc);
}

// batch normalisation XYZZ -> affine: each lane owns `per` consecutive points and shares one
// field inversion among them (Montgomery's trick); zz = 0 stays the infinity encoding (0, 0).
constexpr int kNormPer = 16;
__global__ void __launch_bounds__(256) batch_to_affine_kernel(const void *__restrict__ xyzz, size_t n, void *__restrict__ affine) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t lo = t * kNormPer;
    if (lo >= n) return;
    size_t cnt = n - lo < (size_t)kNormPer ? n - lo : kNormPer;
    FqE prefix[kNormPer];
    FqE run = fe_one<Fq>();
    for (size_t k = 0; k < cnt; k++) {               // den_k = zz * zzz (or 1 for infinity)
        G1Xyzz p = g1_load_xyzz(xyzz, lo + k);
        prefix[k] = run;
        if (!g1_is_inf(p)) run = fq_mul_ni(run, fq_mul_ni(p.zz, p.zzz));
    }
    FqE inv = fq_inv_ni(run);
    for (size_t k = cnt; k-- > 0;) {
        G1Xyzz p = g1_load_xyzz(xyzz, lo + k);
        G1Affine a;
        if (g1_is_inf(p)) {
            a.x = fe_zero<Fq>(); a.y = fe_zero<Fq>();
        } else {
            FqE t_k = fq_mul_ni(inv, prefix[k]);    // 1 / (zz * zzz)
            inv = fq_mul_ni(inv, fq_mul_ni(p.zz, p.zzz));
            a.x = fq_mul_ni(p.x, fq_mul_ni(t_k, p.zzz));
            a.y = fq_mul_ni(p.y, fq_mul_ni(t_k, p.zz));
        }
        g1_store_affine(affine, lo + k, a);
    }
}

// synthetic bases P_i = [a + i d] G: lane t starts at [a + t K d] G and steps by [d] G
__global__ void __launch_bounds__(256) synthetic_bases_kernel(G1Affine g, G1Affine dstep, Fe<Fr381> a_canon, Fe<Fr381> d_canon, size_t n,
                                                              unsigned per, void *__restrict__ out_xyzz) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t lo = t * per;
    if (lo >= n) return;
    // scalar a + lo * d  (mod r not needed for correctness of the group element: plain integer, up to 320 bits)
    uint32_t k[10];
    uint64_t carry = 0;
    uint64_t lo64 = lo;
    uint32_t m0 = (uint32_t)lo64, m1 = (uint32_t)(lo64 >> 32);
    // k = a + d * lo   (schoolbook, d: 8 limbs, lo: 2 limbs)
    uint32_t prod[10];
    for (int i = 0; i < 10; i++) prod[i] = 0;
    for (int i = 0; i < 8; i++) {
        uint64_t c0 = (uint64_t)d_canon.l[i] * m0 + prod[i] + carry;
        prod[i] = (uint32_t)c0;
        carry = c0 >> 32;
    }
    prod[8] = (uint32_t)carry;
    carry = 0;
    for (int i = 0; i < 8; i++) {
        uint64_t c1 = (uint64_t)d_canon.l[i] * m1 + prod[i + 1] + carry;
        prod[i + 1] = (uint32_t)c1;
        carry = c1 >> 32;
    }
    prod[9] = (uint32_t)carry;
    carry = 0;
    for (int i = 0; i < 10; i++) {
        uint64_t s = (uint64_t)prod[i] + (i < 8 ? a_canon.l[i] : 0) + carry;
        k[i] = (uint32_t)s;
        carry = s >> 32;
    }
    G1Xyzz acc = g1_mul_canonical_ni(g, k, 10);
    size_t cnt = n - lo < (size_t)per ? n - lo : per;
    for (size_t j = 0; j < cnt; j++) {
        g1_store_xyzz(out_xyzz, lo + j, acc);
        acc = g1_madd_ni(acc, dstep);
    }
}

}  // namespace zk
