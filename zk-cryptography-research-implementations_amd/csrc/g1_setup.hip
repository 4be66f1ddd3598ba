// g1_setup.hip -- setup-side G1 kernels: fixed-base scalar mul, batch normalisation, pair sums, synthetic bases (own TU).
#include "context.h"
#include "msm_kernels.cuh"

namespace zk {

// out[k] = in[k] + in[k + half]  (affine + affine -> XYZZ): the pre-summed opening bases
// B^(t+1)_k = B^(t)_k + B^(t)_{k + half}  
__global__ void __launch_bounds__(256) g1_pair_add_kernel(const void *__restrict__ in_affine, size_t half, void *__restrict__ out_xyzz) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= half) return;
    g1_store_xyzz(out_xyzz, k, g1_madd(g1_from_affine(g1_load_affine(in_affine, k)), g1_load_affine(in_affine, k + half)));
}

// out[k] = 2^c * in[k]: the window-shifted copies of a small base set (batched opening MSMs, zkmle_kzg.hip)
__global__ void __launch_bounds__(256) g1_shift_kernel(const void *__restrict__ in_affine, size_t n, unsigned c, void *__restrict__ out_xyzz) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    G1Xyzz p = g1_from_affine(g1_load_affine(in_affine, k));
    for (unsigned i = 0; i < c; i++) p = g1_dbl(p);
    g1_store_xyzz(out_xyzz, k, p);
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
        if (v) acc = g1_madd(acc, g1_load_affine(table, (size_t)j * 256 + v));
    }
    g1_store_xyzz(out_xyzz, i, acc);
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
int launch_g1_shift(const void *in_affine, size_t n, unsigned c, void *out_xyzz, hipStream_t s) {
    g1_shift_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(in_affine, n, c, out_xyzz);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int launch_fixed_base_mul(const void *scalars, size_t n, const void *table, void *out_xyzz, hipStream_t s) {
    fixed_base_mul_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(scalars, n, table, out_xyzz);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int launch_batch_to_affine(const void *xyzz, size_t n, void *affine, hipStream_t s) {
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
