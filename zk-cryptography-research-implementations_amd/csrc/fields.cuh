// fields.cuh -- prime-field arithmetic for gfx950 (and the host side of the same library).
//
// Elements are N 32-bit limbs, little-endian, Montgomery form with R = 2^(32 N) = 2^(64 * N/2):
// byte-for-byte the in-memory layout of an arkworks `Fp` (ark-ff 0.5.0, u64 limbs), which is what
// the reference's `Vec<F>` holds (polynomials/src/multilinear/evaluation_form.rs:7-9), so a Rust
// slice can be handed to the C ABI by pointer.  Every result is fully reduced (< p): the
// representation is unique and tables compare bit-for-bit with the CPU reference.
//
// gfx950 notes: 32x32->64 multiply-add is v_mad_u64_u32; the CIOS loops below are fully
// unrolled so all limbs live in VGPRs (no scratch).  No MFMA: this is integer modular arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZK_HD __host__ __device__ __forceinline__

namespace zk {

// ---- field parameter packs (values verified against SURVEY.md Appendix A in tests) ------------
struct Fr381 {
    static constexpr int N = 8;
    static constexpr int ID = 0;
    static ZK_HD uint32_t p(int i) {
        constexpr uint32_t t[N] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return t[i];
    }
    static ZK_HD uint32_t r(int i) {   // R mod p  (Montgomery one)
        constexpr uint32_t t[N] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau, 0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return t[i];
    }
    static ZK_HD uint32_t r2(int i) {  // R^2 mod p
        constexpr uint32_t t[N] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu, 0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return t[i];
    }
    static ZK_HD uint32_t red_k(int i) {   // 2^s R mod p, s = bitlen(p) - 1: mont(hi, red_k) = hi 2^s mod p (mle_kernels.cuh wide_reduce)
        constexpr uint32_t t[N] = {0x7cfca71cu, 0x32667a63u, 0x21e35c08u, 0xc9a97675u, 0xa3ce7067u, 0x67e0272bu, 0xc70c9dbau, 0x58c473f4u};
        return t[i];
    }
    static constexpr uint32_t INV = 0xffffffffu;   // -p^-1 mod 2^32
};
struct Fq381 {
    static constexpr int N = 12;
    static constexpr int ID = 1;
    static ZK_HD uint32_t p(int i) {
        constexpr uint32_t t[N] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return t[i];
    }
    static ZK_HD uint32_t r(int i) {
        constexpr uint32_t t[N] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return t[i];
    }
    static ZK_HD uint32_t r2(int i) {
        constexpr uint32_t t[N] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu, 0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
        return t[i];
    }
    static ZK_HD uint32_t red_k(int i) {   // 2^s R mod p, s = bitlen(p) - 1: mont(hi, red_k) = hi 2^s mod p (mle_kernels.cuh wide_reduce)
        constexpr uint32_t t[N] = {0x41c2f6cau, 0xe20d11f3u, 0x3bc6904eu, 0xeb7dee69u, 0x9ca432ccu, 0x83290cc3u, 0xee4e48a3u, 0x4e671a9eu, 0xd633d08fu, 0xab5fc6a7u, 0x4b7c9801u, 0x17da78abu};
        return t[i];
    }
    static constexpr uint32_t INV = 0xfffcfffdu;
};
struct Bn254Fq {
    static constexpr int N = 8;
    static constexpr int ID = 2;
    static ZK_HD uint32_t p(int i) {
        constexpr uint32_t t[N] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return t[i];
    }
    static ZK_HD uint32_t r(int i) {
        constexpr uint32_t t[N] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return t[i];
    }
    static ZK_HD uint32_t r2(int i) {
        constexpr uint32_t t[N] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
        return t[i];
    }
    static ZK_HD uint32_t red_k(int i) {   // 2^s R mod p, s = bitlen(p) - 1: mont(hi, red_k) = hi 2^s mod p (mle_kernels.cuh wide_reduce)
        constexpr uint32_t t[N] = {0x4580fefau, 0x25e9b10eu, 0x6796d991u, 0x89ad1074u, 0xb1785b0au, 0x1fff6c96u, 0x957d3aa9u, 0x06e79dbcu};
        return t[i];
    }
    static constexpr uint32_t INV = 0xe4866389u;
};
struct Bn254Fr {
    static constexpr int N = 8;
    static constexpr int ID = 3;
    static ZK_HD uint32_t p(int i) {
        constexpr uint32_t t[N] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return t[i];
    }
    static ZK_HD uint32_t r(int i) {
        constexpr uint32_t t[N] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return t[i];
    }
    static ZK_HD uint32_t r2(int i) {
        constexpr uint32_t t[N] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
        return t[i];
    }
    static ZK_HD uint32_t red_k(int i) {   // 2^s R mod p, s = bitlen(p) - 1: mont(hi, red_k) = hi 2^s mod p (mle_kernels.cuh wide_reduce)
        constexpr uint32_t t[N] = {0x33c42db5u, 0x8bf35b7bu, 0x4ba2b94eu, 0x4f86445fu, 0x7aa79b1cu, 0xe893391eu, 0x8c0ffc99u, 0x064f63e4u};
        return t[i];
    }
    static constexpr uint32_t INV = 0xefffffffu;
};

// ---- element ----------------------------------------------------------------------------------
template <class F>
struct alignas(16) Fe {
    uint32_t l[F::N];
};

template <class F> ZK_HD Fe<F> fe_zero() {
    Fe<F> z;
#pragma unroll
    for (int i = 0; i < F::N; i++) z.l[i] = 0;
    return z;
}
template <class F> ZK_HD Fe<F> fe_one() {
    Fe<F> z;
#pragma unroll
    for (int i = 0; i < F::N; i++) z.l[i] = F::r(i);
    return z;
}
template <class F> ZK_HD bool fe_is_zero(const Fe<F> &a) {
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) x |= a.l[i];
    return x == 0;
}
template <class F> ZK_HD bool fe_eq(const Fe<F> &a, const Fe<F> &b) {
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) x |= a.l[i] ^ b.l[i];
    return x == 0;
}

// Carry chains are written with __builtin_addc / __builtin_subc so that hipcc emits v_add_co_u32 / v_addc_co_u32
// (one VALU op per limb); the earlier 64-bit formulation compiled to v_lshl_add_u64 plus operand shuffling, about
// 110 instructions per modular addition (measured in the GKR round kernel).

// r = a - p if a >= p else a   (a < 2p; `hi` is the carry limb above a)
template <class F> ZK_HD void fe_cond_sub_p(Fe<F> &a, uint32_t hi) {
    Fe<F> d;
    unsigned borrow = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) d.l[i] = __builtin_subc(a.l[i], F::p(i), borrow, &borrow);
    // a >= p  <=>  no final borrow, or the carry limb is set
    bool ge = (hi != 0) | (borrow == 0);
#pragma unroll
    for (int i = 0; i < F::N; i++) a.l[i] = ge ? d.l[i] : a.l[i];
}

template <class F> ZK_HD Fe<F> fe_add(const Fe<F> &a, const Fe<F> &b) {
    Fe<F> s;
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) s.l[i] = __builtin_addc(a.l[i], b.l[i], c, &c);
    fe_cond_sub_p<F>(s, c);
    return s;
}

template <class F> ZK_HD Fe<F> fe_sub(const Fe<F> &a, const Fe<F> &b) {
    Fe<F> d;
    unsigned borrow = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) d.l[i] = __builtin_subc(a.l[i], b.l[i], borrow, &borrow);
    // add p back when the subtraction wrapped
    uint32_t mask = (uint32_t)0 - (uint32_t)borrow;
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) d.l[i] = __builtin_addc(d.l[i], F::p(i) & mask, c, &c);
    return d;
}

template <class F> ZK_HD Fe<F> fe_neg(const Fe<F> &a) { return fe_sub<F>(fe_zero<F>(), a); }

template <class F> ZK_HD Fe<F> fe_dbl(const Fe<F> &a) { return fe_add<F>(a, a); }

// Saturated CIOS Montgomery product a*b*R^-1 mod p, fully reduced (host path; on the device see ufield.cuh).  All four moduli leave the top bit of
// the top limb clear, so the running value stays below 2p and t[N] never overflows 32 bits.
template <class F> ZK_HD Fe<F> fe_mul_cios(const Fe<F> &a, const Fe<F> &b) {
    constexpr int N = F::N;
    uint32_t t[N + 1];
#pragma unroll
    for (int i = 0; i <= N; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            c += (uint64_t)a.l[j] * b.l[i] + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        uint32_t tn = t[N] + (uint32_t)c;   // < 2^32: value < 2p*2^32... see note above
        uint32_t m = t[0] * F::INV;
        c = (uint64_t)m * F::p(0) + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            c += (uint64_t)m * F::p(j) + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += tn;
        t[N - 1] = (uint32_t)c;
        t[N] = (uint32_t)(c >> 32);
    }
    Fe<F> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = t[i];
    fe_cond_sub_p<F>(r, t[N]);
    return r;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Host product: the same CIOS on N/2 64-bit limbs with 128-bit accumulators (x86-64 mulx / adcx), ~4x the 32-bit form.  The
// host runs the control path on it: transcript challenges, interpolation, G1 / G2 group laws, the pairing.
template <class F> inline Fe<F> fe_mul_host64(const Fe<F> &a, const Fe<F> &b) {
    constexpr int M = F::N / 2;
    typedef unsigned __int128 u128;
    uint64_t x[M], y[M], p[M], t[M + 2];
    for (int i = 0; i < M; i++) {
        x[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
        y[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
        p[i] = (uint64_t)F::p(2 * i) | ((uint64_t)F::p(2 * i + 1) << 32);
    }
    uint64_t pinv = (uint64_t)(uint32_t)(0u - F::INV);       // p^-1 mod 2^32; one Newton step doubles the precision
    pinv *= 2 - p[0] * pinv;
    const uint64_t inv = 0 - pinv;                           // -p^-1 mod 2^64
    for (int i = 0; i < M + 2; i++) t[i] = 0;
    for (int i = 0; i < M; i++) {
        u128 c = 0;
        for (int j = 0; j < M; j++) {
            c += (u128)x[j] * y[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[M];
        t[M] = (uint64_t)c;
        t[M + 1] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * inv;
        c = (u128)m * p[0] + t[0];
        c >>= 64;
        for (int j = 1; j < M; j++) {
            c += (u128)m * p[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[M];
        t[M - 1] = (uint64_t)c;
        t[M] = t[M + 1] + (uint64_t)(c >> 64);
    }
    Fe<F> r;
    for (int i = 0; i < M; i++) { r.l[2 * i] = (uint32_t)t[i]; r.l[2 * i + 1] = (uint32_t)(t[i] >> 32); }
    fe_cond_sub_p<F>(r, (uint32_t)t[M]);
    return r;
}
#endif

// fe_mul / fe_sqr / fe_inv / canonical conversions are defined in ufield.cuh (the device product is the
// unsaturated 29-bit scan; fe_mul_cios above is the saturated reference form used on the host).

// ---- 16-byte vector load / store of an element (N/4 x dwordx4, coalesced across lanes) ----------
template <class F> __device__ __forceinline__ Fe<F> fe_load(const void *base, size_t idx) {
    const uint4 *p = reinterpret_cast<const uint4 *>(base) + idx * (F::N / 4);
    Fe<F> r;
#pragma unroll
    for (int k = 0; k < F::N / 4; k++) {
        uint4 v = p[k];
        r.l[4 * k + 0] = v.x; r.l[4 * k + 1] = v.y; r.l[4 * k + 2] = v.z; r.l[4 * k + 3] = v.w;
    }
    return r;
}
template <class F> __device__ __forceinline__ void fe_store(void *base, size_t idx, const Fe<F> &a) {
    uint4 *p = reinterpret_cast<uint4 *>(base) + idx * (F::N / 4);
#pragma unroll
    for (int k = 0; k < F::N / 4; k++)
        p[k] = make_uint4(a.l[4 * k + 0], a.l[4 * k + 1], a.l[4 * k + 2], a.l[4 * k + 3]);
}

// runtime field dispatch
#define ZK_DISPATCH_FIELD(field_id, ...)                               \
    switch (field_id) {                                                \
        case 0: { using F = ::zk::Fr381; __VA_ARGS__; } break;         \
        case 1: { using F = ::zk::Fq381; __VA_ARGS__; } break;         \
        case 2: { using F = ::zk::Bn254Fq; __VA_ARGS__; } break;       \
        case 3: { using F = ::zk::Bn254Fr; __VA_ARGS__; } break;       \
        default: return ZK_E_ARG;                                      \
    }

}  // namespace zk
