// ufield.cuh -- unsaturated (29-bit limb) Montgomery multiplication for gfx950.
//
// Why: measured on MI355X (profiles/r1/microbench_instr_rates.jsonl) v_mad_u64_u32 costs 1.8x a plain
// add and every carry-propagating VALU op (v_addc_co_u32, v_lshl_add_u64) 1.7x.  A saturated 32-bit-limb
// CIOS needs carry handling around every multiply-add (hipcc emits ~5 extra VALU ops per v_mad_u64_u32).
// With 29-bit limbs, the 64-bit column accumulators absorb a whole row-scan without ever carrying:
// the inner loops are pure v_mad_u64_u32 chains  T[j] = a[j] * d + T[j]  (addend = the accumulator
// register pair, so no operand shuffling), one carry shift per row.
//
// Representation: L limbs of 29 bits (L = 9 for the 254/255-bit fields, 14 for BLS12-381 Fq), values
// are NOT fully reduced (any representative below ~2^12 p is a valid multiplier input for Fq; below 8 p for
// the 256-bit fields).  Montgomery radix of the scan is 2^(29 L); the stored (arkworks) form uses
// R = 2^(32 N).  umul_std(a_u, b_std) scans the digits of (b << SH), SH = 29 L - 32 N, so it returns
// a * b / 2^(32 N): a drop-in for the saturated product with one operand in each form.
#pragma once
#include "fields.cuh"

namespace zk {

constexpr int UB = 29;
constexpr uint32_t UMASK = (1u << UB) - 1u;

template <class F> struct UParams;
template <> struct UParams<Fr381> {
    static constexpr int L = 9;        // 29-bit limbs
    static constexpr int SH = 5;       // 29 L - 32 N
    static constexpr uint32_t INV = 0x1fffffffu;   // -p^-1 mod 2^29
    static ZK_HD uint32_t p(int i) {
        constexpr uint32_t t[L] = {0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u};
        return t[i];
    }
    static ZK_HD uint32_t p4(int i) {
        constexpr uint32_t t[L] = {0x00000004u, 0x1fffffe0u, 0x1e5bfeffu, 0x0d2017ffu, 0x160154efu, 0x10101343u, 0x1483339du, 0x0994cebeu, 0x01cfb69du};
        return t[i];
    }
    static ZK_HD uint32_t r_u(int i) {
        constexpr uint32_t t[L] = {0x1fffffbau, 0x0000022fu, 0x1cb61180u, 0x0a4e5c00u, 0x0ee8b1a2u, 0x16e6aedfu, 0x1907f8bbu, 0x0853ddf7u, 0x004d043fu};
        return t[i];
    }
    static ZK_HD uint32_t r_std(int i) {
        constexpr uint32_t t[L] = {0x1ffffffeu, 0x0000000fu, 0x00d20080u, 0x096ff400u, 0x04ff5588u, 0x07f7f65eu, 0x15be6631u, 0x0b3598a0u, 0x001824b1u};
        return t[i];
    }
    // uniform-multiplier fold (ufold below): 2 p with every limb but the top one raised by 2^29 (borrowed from the next), 2^261 mod p, 2^290 mod p
    static ZK_HD uint32_t c2p(int i) {
        constexpr uint32_t t[L] = {0x20000002u, 0x3fffffefu, 0x3f2dff7eu, 0x36900bfeu, 0x3b00aa76u, 0x380809a0u, 0x2a4199cdu, 0x34ca675eu, 0x00e7db4du};
        return t[i];
    }
    static ZK_HD uint32_t k7(int i) {
        constexpr uint32_t t[L] = {0x1fffffbau, 0x0000022fu, 0x1cb61180u, 0x0a4e5c00u, 0x0ee8b1a2u, 0x16e6aedfu, 0x1907f8bbu, 0x0853ddf7u, 0x004d043fu};
        return t[i];
    }
    static ZK_HD uint32_t k8(int i) {
        constexpr uint32_t t[L] = {0x0abdac49u, 0x0a129d71u, 0x06a3eff5u, 0x168d894du, 0x15997df8u, 0x09407325u, 0x0b7bc5dcu, 0x1ec6f83eu, 0x0071e0beu};
        return t[i];
    }
};

template <> struct UParams<Fq381> {
    static constexpr int L = 14;        // 29-bit limbs
    static constexpr int SH = 22;       // 29 L - 32 N
    static constexpr uint32_t INV = 0x1ffcfffdu;   // -p^-1 mod 2^29
    static ZK_HD uint32_t p(int i) {
        constexpr uint32_t t[L] = {0x1fffaaabu, 0x0ff7ffffu, 0x14ffffeeu, 0x17fffd62u, 0x0f6241eau, 0x09507b58u, 0x0afd9cc3u, 0x109e70a2u, 0x1764774bu, 0x121a5d66u, 0x12c6e9edu, 0x12ffcd34u, 0x00111ea3u, 0x0000000du};
        return t[i];
    }
    static ZK_HD uint32_t p4(int i) {
        constexpr uint32_t t[L] = {0x1ffeaaacu, 0x1fdfffffu, 0x13ffffb9u, 0x1ffff58au, 0x1d8907aau, 0x0541ed61u, 0x0bf6730du, 0x0279c289u, 0x1d91dd2eu, 0x0869759au, 0x0b1ba7b6u, 0x0bff34d2u, 0x00447a8eu, 0x00000034u};
        return t[i];
    }
    static ZK_HD uint32_t r_u(int i) {
        constexpr uint32_t t[L] = {0x03a9fb84u, 0x0ba00690u, 0x071288f1u, 0x0f59bcc5u, 0x126cb614u, 0x0585bf36u, 0x1b85ac3du, 0x1cf856fau, 0x1891ecbdu, 0x1a7eec05u, 0x155a88f0u, 0x0741ac6du, 0x1317c30fu, 0x00000009u};
        return t[i];
    }
    static ZK_HD uint32_t r_std(int i) {
        constexpr uint32_t t[L] = {0x0002fffdu, 0x10480000u, 0x0300009du, 0x08001788u, 0x158baebfu, 0x0c2ba9e3u, 0x1d157d22u, 0x0a6e0a4au, 0x0d77ce58u, 0x1d12b763u, 0x1701c6a5u, 0x1501c926u, 0x1f65ec3fu, 0x0000000au};
        return t[i];
    }
};

template <> struct UParams<Bn254Fq> {
    static constexpr int L = 9;        // 29-bit limbs
    static constexpr int SH = 5;       // 29 L - 32 N
    static constexpr uint32_t INV = 0x04866389u;   // -p^-1 mod 2^29
    static ZK_HD uint32_t p(int i) {
        constexpr uint32_t t[L] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
        return t[i];
    }
    static ZK_HD uint32_t p4(int i) {
        constexpr uint32_t t[L] = {0x01f3f51cu, 0x041182dbu, 0x11ca8d3cu, 0x0b548b43u, 0x161765e0u, 0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u};
        return t[i];
    }
    static ZK_HD uint32_t r_u(int i) {
        constexpr uint32_t t[L] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
        return t[i];
    }
    static ZK_HD uint32_t r_std(int i) {
        constexpr uint32_t t[L] = {0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
        return t[i];
    }
    // uniform-multiplier fold (ufold below): 2 p with every limb but the top one raised by 2^29 (borrowed from the next), 2^261 mod p, 2^290 mod p
    static ZK_HD uint32_t c2p(int i) {
        constexpr uint32_t t[L] = {0x30f9fa8eu, 0x2208c16cu, 0x38e5469du, 0x25aa45a0u, 0x2b0bb2efu, 0x25b68180u, 0x214dc281u, 0x3cb84c67u, 0x0060c89bu};
        return t[i];
    }
    static ZK_HD uint32_t k7(int i) {
        constexpr uint32_t t[L] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
        return t[i];
    }
    static ZK_HD uint32_t k8(int i) {
        constexpr uint32_t t[L] = {0x0dfff5d8u, 0x103ae4bdu, 0x097ea07du, 0x04d7e9a4u, 0x12091fbau, 0x0e81c6b7u, 0x06e1271bu, 0x1f763611u, 0x001af00du};
        return t[i];
    }
};

template <> struct UParams<Bn254Fr> {
    static constexpr int L = 9;        // 29-bit limbs
    static constexpr int SH = 5;       // 29 L - 32 N
    static constexpr uint32_t INV = 0x0fffffffu;   // -p^-1 mod 2^29
    static ZK_HD uint32_t p(int i) {
        constexpr uint32_t t[L] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
        return t[i];
    }
    static ZK_HD uint32_t p4(int i) {
        constexpr uint32_t t[L] = {0x00000004u, 0x1c3eb27eu, 0x19709143u, 0x1f4243cdu, 0x16174a0cu, 0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u};
        return t[i];
    }
    static ZK_HD uint32_t r_u(int i) {
        constexpr uint32_t t[L] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
        return t[i];
    }
    static ZK_HD uint32_t r_std(int i) {
        constexpr uint32_t t[L] = {0x0ffffffbu, 0x04b1a0e2u, 0x18334a6bu, 0x18ed2b3eu, 0x1462e36fu, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
        return t[i];
    }
    // uniform-multiplier fold (ufold below): 2 p with every limb but the top one raised by 2^29 (borrowed from the next), 2^261 mod p, 2^290 mod p
    static ZK_HD uint32_t c2p(int i) {
        constexpr uint32_t t[L] = {0x20000002u, 0x3e1f593eu, 0x3cb848a0u, 0x2fa121e5u, 0x2b0ba505u, 0x25b68180u, 0x214dc281u, 0x3cb84c67u, 0x0060c89bu};
        return t[i];
    }
    static ZK_HD uint32_t k7(int i) {
        constexpr uint32_t t[L] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
        return t[i];
    }
    static ZK_HD uint32_t k8(int i) {
        constexpr uint32_t t[L] = {0x16e2e768u, 0x1cf10ca2u, 0x1ea15493u, 0x19213582u, 0x0e0e45b7u, 0x0e866055u, 0x06e1271bu, 0x1f763611u, 0x001af00du};
        return t[i];
    }
};

template <class F> struct Ufe {
    uint32_t l[UParams<F>::L];
};

// ---- conversions ----------------------------------------------------------------------------------------
// 32-bit limbs -> 29-bit limbs of the same integer (value < 2^(32 N) <= 2^(29 L))
template <class F> ZK_HD Ufe<F> u_from_limbs32(const Fe<F> &a) {
    constexpr int L = UParams<F>::L, N = F::N;
    Ufe<F> r;
#pragma unroll
    for (int i = 0; i < L; i++) {
        int q = UB * i, w = q >> 5, off = q & 31;
        uint32_t lo = w < N ? a.l[w] : 0u, hi = (w + 1) < N ? a.l[w + 1] : 0u;
        uint32_t v = off ? ((lo >> off) | (hi << (32 - off))) : lo;
        r.l[i] = v & UMASK;
    }
    return r;
}
// normalized 29-bit limbs (< 2^(32 N)) -> 32-bit limbs
template <class F> ZK_HD Fe<F> u_to_limbs32(const Ufe<F> &a) {
    constexpr int L = UParams<F>::L, N = F::N;
    Fe<F> r;
#pragma unroll
    for (int w = 0; w < N; w++) {
        int q = 32 * w, i = q / UB, off = q - UB * i;      // bit q lives in limb i at offset off
        uint64_t v = (uint64_t)a.l[i] >> off;
        int have = UB - off;
        if (i + 1 < L) v |= (uint64_t)a.l[i + 1] << have;
        if (have + UB < 32 && i + 2 < L) v |= (uint64_t)a.l[i + 2] << (have + UB);
        r.l[w] = (uint32_t)v;
    }
    return r;
}
// digit i (29 bits) of (b << SH), b in 32-bit limbs
template <class F> ZK_HD uint32_t u_digit_std(const Fe<F> &b, int i) {
    constexpr int N = F::N, SH = UParams<F>::SH;
    int q = UB * i - SH;
    if (q < 0) return (b.l[0] << (-q)) & UMASK;
    int w = q >> 5, off = q & 31;
    uint32_t lo = w < N ? b.l[w] : 0u, hi = (w + 1) < N ? b.l[w + 1] : 0u;
    uint32_t v = off ? ((lo >> off) | (hi << (32 - off))) : lo;
    return v & UMASK;
}

// ---- the row scan -------------------------------------------------------------------------------------------
// T (L 64-bit columns) += a * d ; one Montgomery step with radix 2^29 ; shift one limb down
template <class F> ZK_HD void u_row(uint64_t (&T)[UParams<F>::L], const Ufe<F> &a, uint32_t d) {
    constexpr int L = UParams<F>::L;
#pragma unroll
    for (int j = 0; j < L; j++) T[j] += (uint64_t)a.l[j] * d;
    uint32_t m = ((uint32_t)T[0] * UParams<F>::INV) & UMASK;
#pragma unroll
    for (int j = 0; j < L; j++) T[j] += (uint64_t)m * UParams<F>::p(j);
    uint64_t carry = T[0] >> UB;                       // the low 29 bits of T[0] are now zero
#pragma unroll
    for (int j = 0; j + 1 < L; j++) T[j] = T[j + 1];
    T[L - 1] = 0;
    T[0] += carry;
}
template <class F> ZK_HD Ufe<F> u_normalize_columns(uint64_t (&T)[UParams<F>::L]) {
    constexpr int L = UParams<F>::L;
    Ufe<F> r;
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < L; j++) {
        uint64_t v = T[j] + c;
        r.l[j] = (uint32_t)v & UMASK;
        c = v >> UB;
    }
    return r;   // value < 2^(29 L): the final carry is zero
}

// a (29-bit form) * b (stored 32-bit Montgomery form) / 2^(32 N)  mod p ; result < 2 p, normalized limbs
template <class F> ZK_HD Ufe<F> umul_std(const Ufe<F> &a, const Fe<F> &b) {
    constexpr int L = UParams<F>::L;
    uint64_t T[L];
#pragma unroll
    for (int j = 0; j < L; j++) T[j] = 0;
#pragma unroll
    for (int i = 0; i < L; i++) u_row<F>(T, a, u_digit_std<F>(b, i));
    return u_normalize_columns<F>(T);
}
// a * b / 2^(29 L) mod p, both in 29-bit form (b normalized: limbs < 2^29; a limbs < 2^30)
template <class F> ZK_HD Ufe<F> umul(const Ufe<F> &a, const Ufe<F> &b) {
    constexpr int L = UParams<F>::L;
    uint64_t T[L];
#pragma unroll
    for (int j = 0; j < L; j++) T[j] = 0;
#pragma unroll
    for (int i = 0; i < L; i++) u_row<F>(T, a, b.l[i]);
    return u_normalize_columns<F>(T);
}
// (a1 * b1 + a2 * b2) / 2^(29 L) mod p with ONE Montgomery reduction per row: 3 L^2 multiply-adds instead of 4 L^2 for two
// products.  All four operands normalized (limbs < 2^29): a live column then collects at most L * 3 * 2^58 < 2^64.
template <class F> ZK_HD Ufe<F> umul2(const Ufe<F> &a1, const Ufe<F> &b1, const Ufe<F> &a2, const Ufe<F> &b2) {
    constexpr int L = UParams<F>::L;
    uint64_t T[L];
#pragma unroll
    for (int j = 0; j < L; j++) T[j] = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
#pragma unroll
        for (int j = 0; j < L; j++) T[j] += (uint64_t)a1.l[j] * b1.l[i];
        u_row<F>(T, a2, b2.l[i]);
    }
    return u_normalize_columns<F>(T);
}
// (a1 * b1 + a2 * b2) / 2^(32 N) mod p for stored-form right operands (the digits of b << SH are scanned, as in umul_std): the sum of two
// stored-form products as ONE scan with one Montgomery reduction per row.  For canonical operands (all four < p) the result is below
// 2 p^2 / 2^(32 N) + p < 2 p (p < 2^(32 N - 1) for the four moduli), so one conditional subtraction finishes it.
template <class F> ZK_HD Ufe<F> umul2_std(const Ufe<F> &a1, const Fe<F> &b1, const Ufe<F> &a2, const Fe<F> &b2) {
    constexpr int L = UParams<F>::L;
    uint64_t T[L];
#pragma unroll
    for (int j = 0; j < L; j++) T[j] = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
        const uint32_t d1 = u_digit_std<F>(b1, i);
#pragma unroll
        for (int j = 0; j < L; j++) T[j] += (uint64_t)a1.l[j] * d1;
        u_row<F>(T, a2, u_digit_std<F>(b2, i));
    }
    return u_normalize_columns<F>(T);
}
template <class F> ZK_HD Ufe<F> u_reduce_once(const Ufe<F> &a);      // below
// a1 * b1 + a2 * b2 in the stored form, canonical: bit-identical to fe_add(fe_mul(a1, b1), fe_mul(a2, b2))
template <class F> ZK_HD Fe<F> fe_mul2_u(const Fe<F> &a1, const Fe<F> &b1, const Fe<F> &a2, const Fe<F> &b2) {
    return u_to_limbs32<F>(u_reduce_once<F>(umul2_std<F>(u_from_limbs32<F>(a1), b1, u_from_limbs32<F>(a2), b2)));
}
// a^2 / 2^(29 L): the row scan with the symmetric products taken once.  At step t the live column j holds
// absolute weight j + t, so a_j a_t (j > t) is added doubled at step t only; every contribution to the column
// that the Montgomery step consumes (weight t) comes from steps <= t/2, so T[0] is complete when it is used.
// L (L + 1) / 2 multiply-adds instead of L^2 for the a * a half.
template <class F> ZK_HD Ufe<F> usqr(const Ufe<F> &a) {
    constexpr int L = UParams<F>::L;
    uint64_t T[L];
#pragma unroll
    for (int j = 0; j < L; j++) T[j] = 0;
#pragma unroll
    for (int t = 0; t < L; t++) {
        uint32_t at = a.l[t], at2 = at << 1;
        T[t] += (uint64_t)at * at;
#pragma unroll
        for (int j = t + 1; j < L; j++) T[j] += (uint64_t)a.l[j] * at2;
        uint32_t m = ((uint32_t)T[0] * UParams<F>::INV) & UMASK;
#pragma unroll
        for (int j = 0; j < L; j++) T[j] += (uint64_t)m * UParams<F>::p(j);
        uint64_t carry = T[0] >> UB;
#pragma unroll
        for (int j = 0; j + 1 < L; j++) T[j] = T[j + 1];
        T[L - 1] = 0;
        T[0] += carry;
    }
    return u_normalize_columns<F>(T);
}

// ---- additive operations on unreduced values ----------------------------------------------------------------
template <class F> ZK_HD Ufe<F> u_zero() {
    Ufe<F> r;
#pragma unroll
    for (int j = 0; j < UParams<F>::L; j++) r.l[j] = 0;
    return r;
}
template <class F> ZK_HD bool u_is_exact_zero(const Ufe<F> &a) {
    uint32_t x = 0;
#pragma unroll
    for (int j = 0; j < UParams<F>::L; j++) x |= a.l[j];
    return x == 0;
}
// a + b, limbs renormalized (no modular reduction)
template <class F> ZK_HD Ufe<F> uadd(const Ufe<F> &a, const Ufe<F> &b) {
    constexpr int L = UParams<F>::L;
    Ufe<F> r;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < L; j++) {
        uint32_t v = a.l[j] + b.l[j] + c;
        r.l[j] = j + 1 < L ? (v & UMASK) : v;
        c = v >> UB;
    }
    return r;
}
// a - b + 4 p  for b < 4 p (every subtrahend on the G1 path is a product (< 2 p) or a doubled product)
template <class F> ZK_HD Ufe<F> usub(const Ufe<F> &a, const Ufe<F> &b) {
    constexpr int L = UParams<F>::L;
    Ufe<F> r;
    int32_t c = 0;
#pragma unroll
    for (int j = 0; j < L; j++) {
        int32_t v = (int32_t)a.l[j] + (int32_t)UParams<F>::p4(j) - (int32_t)b.l[j] + c;
        r.l[j] = j + 1 < L ? ((uint32_t)v & UMASK) : (uint32_t)v;
        c = v >> UB;                                   // arithmetic shift: signed borrow
    }
    return r;
}
// value in [0, 2 p), normalized -> fully reduced (conditional subtraction of p)
template <class F> ZK_HD Ufe<F> u_reduce_once(const Ufe<F> &a) {
    constexpr int L = UParams<F>::L;
    Ufe<F> d;
    int32_t c = 0;
#pragma unroll
    for (int j = 0; j < L; j++) {
        int32_t v = (int32_t)a.l[j] - (int32_t)UParams<F>::p(j) + c;
        d.l[j] = (uint32_t)v & UMASK;
        c = v >> UB;
    }
    bool ge = c >= 0;                                  // no final borrow: a >= p
    Ufe<F> r;
#pragma unroll
    for (int j = 0; j < L; j++) r.l[j] = ge ? d.l[j] : a.l[j];
    return r;
}
// is a product-range value (< 2 p) congruent to 0, i.e. equal to 0 or p
template <class F> ZK_HD bool u_is_zero_mod_p(const Ufe<F> &a) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int j = 0; j < UParams<F>::L; j++) { z |= a.l[j]; e |= a.l[j] ^ UParams<F>::p(j); }
    return z == 0 || e == 0;
}

// ---- drop-in saturated product through the unsaturated scan --------------------------------------------------
// a * b / R mod p for canonical stored-form inputs, canonical output: bit-identical to fe_mul
template <class F> ZK_HD Fe<F> fe_mul_u(const Fe<F> &a, const Fe<F> &b) {
    return u_to_limbs32<F>(u_reduce_once<F>(umul_std<F>(u_from_limbs32<F>(a), b)));
}
// with the left operand already converted (loop-invariant multiplier, e.g. the fold challenge)
template <class F> ZK_HD Fe<F> fe_mul_u_pre(const Ufe<F> &a_u, const Fe<F> &b) {
    return u_to_limbs32<F>(u_reduce_once<F>(umul_std<F>(a_u, b)));
}

// stored Montgomery form (R = 2^(32 N)) <-> internal form (radix 2^(29 L)), values < 2 p
template <class F> ZK_HD Ufe<F> u_from_std(const Fe<F> &x) {        // x R_std -> x R_u : multiply by (R_u as an integer)
    Ufe<F> ru;
#pragma unroll
    for (int j = 0; j < UParams<F>::L; j++) ru.l[j] = UParams<F>::r_u(j);
    return umul_std<F>(ru, x);
}
template <class F> ZK_HD Fe<F> u_to_std(const Ufe<F> &x) {          // x R_u -> x R_std, canonical
    Ufe<F> rs;
#pragma unroll
    for (int j = 0; j < UParams<F>::L; j++) rs.l[j] = UParams<F>::r_std(j);
    return u_to_limbs32<F>(u_reduce_once<F>(umul<F>(x, rs)));
}


// ---- a UNIFORM multiplier: out = a + r (b - a) with r the same for every lane (a round's challenge) ---------------
// A Montgomery product spends L^2 multiply-adds on the product and L^2 on its reduction.  When the multiplier r is the same for
// every element of a pass, its shifted multiples  R_i = r 2^(29 (i + 2)) mod p,  i < L,  are L constants of the pass (81 words for the
// 9-limb fields: scalar registers), and  r D 2^58 = sum_i d_i R_i  (mod p)  for the 29-bit digits d_i of D: L^2 multiply-adds into L
// columns with no dependency between rows, a value below L 2^31 p -- 34 bits too long -- which TWO Montgomery rows (2 L multiply-adds,
// a division by 2^58) bring below p (1 + 2^-24).  99 multiply-adds where the scan takes 162 (r4).
//   D = b - a + 2 p, limb by limb with no carries: the constant is 2 p with every limb but the top raised by 2^29 (c2p), so that every
//   digit stays positive and below 2^31; a 2^58 rides along in the columns (a_j enters column j + 2), which makes the sum
//   a + r (b - a) come out of the same two rows.  Inputs: normalized limbs of values below 2^(29 L - 5) (canonical elements here).
//   Output: normalized limbs of a value congruent to a + r (b - a), below a + p (1 + 2^-24).
template <class F> struct UniMul {
    uint32_t t[UParams<F>::L][UParams<F>::L];               // t[i] = r 2^(29 (i + 2)) mod p, canonical, 29-bit limbs
};
// row i of the table from the challenge in the stored form (r R_std mod p): (2^(29 (i + 2)) mod p) r R_std / R_std
template <class F> ZK_HD Ufe<F> unimul_row(const Fe<F> &r_std, int i) {
    constexpr int L = UParams<F>::L;
    Ufe<F> k;
#pragma unroll
    for (int j = 0; j < L; j++) k.l[j] = i == L - 2 ? UParams<F>::k7(j) : i == L - 1 ? UParams<F>::k8(j) : (j == i + 2 ? 1u : 0u);
    return u_reduce_once<F>(umul_std<F>(k, r_std));
}
template <class F> ZK_HD void unimul_from(UniMul<F> &m, const Fe<F> &r_std) {
    for (int i = 0; i < UParams<F>::L; i++) {
        const Ufe<F> row = unimul_row<F>(r_std, i);
        for (int j = 0; j < UParams<F>::L; j++) m.t[i][j] = row.l[j];
    }
}
template <class F> ZK_HD Ufe<F> ufold(const UniMul<F> &m, const Ufe<F> &a, const Ufe<F> &b) {
    constexpr int L = UParams<F>::L;
    uint32_t d[L];
#pragma unroll
    for (int j = 0; j < L; j++) d[j] = b.l[j] - a.l[j] + UParams<F>::c2p(j);
    uint64_t T[L];
    T[0] = 0;
    T[1] = 0;
#pragma unroll
    for (int j = 2; j < L; j++) T[j] = a.l[j - 2];
#pragma unroll
    for (int i = 0; i < L; i++) {
#pragma unroll
        for (int j = 0; j < L; j++) T[j] += (uint64_t)d[i] * m.t[i][j];
    }
#pragma unroll
    for (int row = 0; row < 2; row++) {
        const uint32_t q = ((uint32_t)T[0] * UParams<F>::INV) & UMASK;
#pragma unroll
        for (int j = 0; j < L; j++) T[j] += (uint64_t)q * UParams<F>::p(j);
        const uint64_t carry = T[0] >> UB;
#pragma unroll
        for (int j = 0; j + 1 < L; j++) T[j] = T[j + 1];
        T[L - 1] = a.l[L - 2 + row];
        T[0] += carry;
    }
    return u_normalize_columns<F>(T);
}
// normalized limbs of a value below 2 p (1 + 2^-24) (what ufold leaves for canonical inputs) -> the canonical element, 32-bit limbs.
// The second subtraction is taken by one lane in 2^31 or so (value >= 2 p: both a and the product within 2^-24 p of p).
template <class F> ZK_HD Fe<F> fe_from_u_below_2p(const Ufe<F> &x) {
    Fe<F> s = u_to_limbs32<F>(x);
    fe_cond_sub_p<F>(s, 0);
    if (s.l[F::N - 1] >= F::p(F::N - 1)) fe_cond_sub_p<F>(s, 0);
    return s;
}

// ---- the library's field product --------------------------------------------------------------------------------
// device: unsaturated scan (1.45x the saturated CIOS as hipcc compiles it, 1.9x without the conversions);
// host:   saturated CIOS on 64-bit limbs (fields.cuh).  Both are fully reduced, hence bit-identical.
template <class F> ZK_HD Fe<F> fe_mul(const Fe<F> &a, const Fe<F> &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fe_mul_u<F>(a, b);
#else
    return fe_mul_host64<F>(a, b);
#endif
}

template <class F> ZK_HD Fe<F> fe_sqr(const Fe<F> &a) { return fe_mul<F>(a, a); }

// Montgomery form -> canonical integer (into_bigint()): multiply by the raw integer 1
template <class F> ZK_HD Fe<F> fe_to_canonical(const Fe<F> &a) {
    Fe<F> one = fe_zero<F>();
    one.l[0] = 1;
    return fe_mul<F>(a, one);
}
template <class F> ZK_HD Fe<F> fe_from_canonical(const Fe<F> &c) {
    Fe<F> r2;
#pragma unroll
    for (int i = 0; i < F::N; i++) r2.l[i] = F::r2(i);
    return fe_mul<F>(c, r2);
}
template <class F> ZK_HD Fe<F> fe_from_u64(uint64_t v) {
    Fe<F> c = fe_zero<F>();
    c.l[0] = (uint32_t)v;
    c.l[1] = (uint32_t)(v >> 32);
    return fe_from_canonical<F>(c);
}

// a^(p-2); host-side helper (Lagrange interpolation, batch normalisation)
template <class F> ZK_HD Fe<F> fe_inv(const Fe<F> &a) {
    Fe<F> acc = fe_one<F>(), base = a;
    uint32_t borrow = 2;   // exponent p - 2, computed limb by limb
    for (int i = 0; i < F::N; i++) {
        uint32_t pi = F::p(i);
        uint32_t e = pi - borrow;
        borrow = (pi < borrow) ? 1u : 0u;
        for (int k = 0; k < 32; k++) {
            if ((e >> k) & 1) acc = fe_mul<F>(acc, base);
            base = fe_sqr<F>(base);
        }
    }
    return acc;
}

}  // namespace zk
