// g1_setup.hip -- setup-side G1 kernels: fixed-base scalar mul, batch normalisation, pair sums, synthetic bases (own TU).
#include "context.h"
#include "msm_kernels.cuh"
#include "g1u.cuh"

namespace zk {

// out[k] = in[k] + in[k + half]  (affine + affine -> XYZZ): the pre-summed opening bases
// B^(t+1)_k = B^(t)_k + B^(t)_{k + half}  
__global__ void __launch_bounds__(256) g1_pair_add_kernel(const void *__restrict__ in_affine, size_t half, void *__restrict__ out_xyzz) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= half) return;
    g1_store_xyzz(out_xyzz, k, g1_madd(g1_from_affine(g1_load_affine(in_affine, k)), g1_load_affine(in_affine, k + half)));
}

// the same on stored XYZZ input (the short levels are summed in a chain and normalised together at the end)
__global__ void __launch_bounds__(256) g1_pair_add_xyzz_kernel(const void *__restrict__ in_xyzz, size_t half, void *__restrict__ out_xyzz) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= half) return;
    g1_store_xyzz(out_xyzz, k, g1_add(g1_load_xyzz(in_xyzz, k), g1_load_xyzz(in_xyzz, k + half)));
}

// out[k] = 2^c * in[k]: the window-shifted copies of a small base set (batched opening MSMs, zkmle_kzg.hip).  The input is
// stored affine (`in_is_xyzz` = 0) or stored XYZZ (a previous shift, not normalised in between); the doublings run in the
// internal form.
__global__ void __launch_bounds__(256) g1_shift_kernel(const void *__restrict__ in, int in_is_xyzz, size_t n, unsigned c, void *__restrict__ out_xyzz) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    G1Xyzz p = in_is_xyzz ? g1_load_xyzz(in, k) : g1_from_affine(g1_load_affine(in, k));
    if (g1_is_inf(p)) { g1_store_xyzz(out_xyzz, k, g1_xyzz_inf()); return; }
    G1XyzzU q;
    q.x = u_from_std<Fq381>(p.x); q.y = u_from_std<Fq381>(p.y); q.zz = u_from_std<Fq381>(p.zz); q.zzz = u_from_std<Fq381>(p.zzz);
    q.inf = false;
    for (unsigned i = 0; i < c; i++) q = g1u_dbl(q);
    g1_store_xyzz(out_xyzz, k, g1u_to_std(q));
}

// fixed-base scalar multiplication  out[i] = [s_i] G  with a 16-bit-window table of G in the internal form
// (table16[j * 65536 + v] = [v * 65536^j] G, 16 x 65536 pre-converted affine points, 128 MiB): 16 mixed additions per point
// (compute_g1_powers_of_tau trusted_setup.rs:51-60 does one 255-bit double-and-add per point; r1 started with a byte-window
// table and 32 additions per point: 88 ms for 2^24 points).
__global__ void __launch_bounds__(256) fixed_base_mul_kernel(const void *__restrict__ scalars, size_t n, const void *__restrict__ table16,
                                                             void *__restrict__ out_xyzz) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<Fr381> k = fe_to_canonical<Fr381>(fe_load<Fr381>(scalars, i));
    G1XyzzU acc = g1u_inf();
    for (int j = 0; j < 16; j++) {
        unsigned v = (k.l[j >> 1] >> (16 * (j & 1))) & 0xffffu;
        if (v) g1u_madd(acc, g1u_load_affine(table16, (size_t)j * 65536 + v), false);
    }
    g1_store_xyzz(out_xyzz, i, g1u_to_std(acc));
}
// the 16-bit-window table from the byte-window table (table8[j * 256 + v] = [v * 256^j] G, stored affine, (0, 0) = infinity):
// entry (j, v) = table8[2 j][v & 255] + table8[2 j + 1][v >> 8]
__global__ void __launch_bounds__(256) fixed_table16_kernel(const void *__restrict__ table8, void *__restrict__ out_xyzz) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)16 * 65536) return;
    unsigned j = (unsigned)(id >> 16), v = (unsigned)(id & 0xffffu), lo = v & 255u, hi = v >> 8;
    G1Xyzz acc = g1_xyzz_inf();
    if (lo) acc = g1_from_affine(g1_load_affine(table8, (size_t)(2 * j) * 256 + lo));
    if (hi) {
        G1Affine q = g1_load_affine(table8, (size_t)(2 * j + 1) * 256 + hi);
        acc = lo ? g1_madd(acc, q) : g1_from_affine(q);
    }
    g1_store_xyzz(out_xyzz, id, acc);
}

// batch normalisation XYZZ -> affine: each lane owns `per` consecutive points and shares one
// field inversion among them (Montgomery's trick); zz = 0 stays the infinity encoding (0, 0).
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
        if (!g1_is_inf(p)) run = fe_mul<Fq>(run, fe_mul<Fq>(p.zz, p.zzz));
    }
    FqE inv = fe_inv<Fq>(run);
    for (size_t k = cnt; k-- > 0;) {
        G1Xyzz p = g1_load_xyzz(xyzz, lo + k);
        G1Affine a;
        if (g1_is_inf(p)) {
            a.x = fe_zero<Fq>(); a.y = fe_zero<Fq>();
        } else {
            FqE t_k = fe_mul<Fq>(inv, prefix[k]);    // 1 / (zz * zzz)
            inv = fe_mul<Fq>(inv, fe_mul<Fq>(p.zz, p.zzz));
            a.x = fe_mul<Fq>(p.x, fe_mul<Fq>(t_k, p.zzz));
            a.y = fe_mul<Fq>(p.y, fe_mul<Fq>(t_k, p.zz));
        }
        g1_store_affine(affine, lo + k, a);
    }
}

// the same for large arrays: lane t owns the points t, t + T, t + 2 T, ... (coalesced across lanes), keeps its running prefix
// products in `scratch` (one Fq element per point) instead of registers, and so shares one inversion among `n / T` points
// (64 instead of 16: the inversion is ~570 products, more than the 9 useful ones per point of the short batches).
__global__ void __launch_bounds__(256) batch_to_affine_strided_kernel(const void *__restrict__ xyzz, size_t n, size_t T, void *__restrict__ scratch,
                                                                      void *__restrict__ affine) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T || t >= n) return;
    FqE run = fe_one<Fq>();
    for (size_t i = t; i < n; i += T) {
        G1Xyzz p = g1_load_xyzz(xyzz, i);
        fe_store<Fq>(scratch, i, run);
        if (!g1_is_inf(p)) run = fe_mul<Fq>(run, fe_mul<Fq>(p.zz, p.zzz));
    }
    FqE inv = fe_inv<Fq>(run);
    for (size_t k = (n - t + T - 1) / T; k-- > 0;) {
        size_t i = t + k * T;
        G1Xyzz p = g1_load_xyzz(xyzz, i);
        G1Affine a;
        if (g1_is_inf(p)) {
            a.x = fe_zero<Fq>(); a.y = fe_zero<Fq>();
        } else {
            FqE t_k = fe_mul<Fq>(inv, fe_load<Fq>(scratch, i));     // 1 / (zz * zzz)
            inv = fe_mul<Fq>(inv, fe_mul<Fq>(p.zz, p.zzz));
            a.x = fe_mul<Fq>(p.x, fe_mul<Fq>(t_k, p.zzz));
            a.y = fe_mul<Fq>(p.y, fe_mul<Fq>(t_k, p.zz));
        }
        g1_store_affine(affine, i, a);
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
    G1Xyzz acc = g1_mul_canonical(g, k, 10);
    size_t cnt = n - lo < (size_t)per ? n - lo : per;
    for (size_t j = 0; j < cnt; j++) {
        g1_store_xyzz(out_xyzz, lo + j, acc);
        acc = g1_madd(acc, dstep);
    }
}


int launch_g1_pair_add(const void *in_affine, size_t half, void *out_xyzz, hipStream_t s) {
    g1_pair_add_kernel<<<(unsigned)((half + 255) / 256), 256, 0, s>>>(in_affine, half, out_xyzz);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int launch_g1_pair_add_xyzz(const void *in_xyzz, size_t half, void *out_xyzz, hipStream_t s) {
    g1_pair_add_xyzz_kernel<<<(unsigned)((half + 255) / 256), 256, 0, s>>>(in_xyzz, half, out_xyzz);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int launch_g1_shift(const void *in, int in_is_xyzz, size_t n, unsigned c, void *out_xyzz, hipStream_t s) {
    g1_shift_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(in, in_is_xyzz, n, c, out_xyzz);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int launch_fixed_base_mul(const void *scalars, size_t n, const void *table16, void *out_xyzz, hipStream_t s) {
    fixed_base_mul_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(scalars, n, table16, out_xyzz);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int launch_fixed_table16(const void *table8, void *out_xyzz, hipStream_t s) {
    fixed_table16_kernel<<<16 * 65536 / 256, 256, 0, s>>>(table8, out_xyzz);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int launch_batch_to_affine(const void *xyzz, size_t n, void *affine, hipStream_t s) {
    if (n >= ((size_t)1 << 21)) {                       // long batches, prefix products in pooled scratch (reused in stream order)
        size_t per = n >> 16;
        if (per > 64) per = 64;
        const size_t T = (n + per - 1) / per;
        void *scratch = nullptr;
        ZK_TRY(pool_alloc(n * sizeof(FqE), &scratch));
        batch_to_affine_strided_kernel<<<(unsigned)((T + 255) / 256), 256, 0, s>>>(xyzz, n, T, scratch, affine);
        hipError_t e = hipGetLastError();
        pool_free(scratch);
        ZK_HIP(e);
        return ZK_OK;
    }
    size_t threads = (n + kNormPer - 1) / kNormPer;
    batch_to_affine_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(xyzz, n, affine);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int launch_synthetic_bases(const G1Affine &g, const G1Affine &dstep, const Fe<Fr381> &a_canon, const Fe<Fr381> &d_canon, size_t n,
                           unsigned per, void *out_xyzz, hipStream_t s) {
    size_t threads = (n + per - 1) / per;
    synthetic_bases_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(g, dstep, a_canon, d_canon, n, per, out_xyzz);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

}  // namespace zk
