// g1.cuh -- BLS12-381 G1 (y^2 = x^3 + 4 over Fq) point arithmetic for gfx950 and the host.
//
// The reference does its group arithmetic in ark-ec 0.5.0 (`P::G1` projective, mul_bigint, Sum:
// multilinear_kzg/src/multilinear_kzg.rs:37-42, trusted_setup.rs:51-60) [ext].  Only the GROUP
// ELEMENT is observable (results are compared in normalised affine form), so the coordinate
// system is ours: bases are affine (x, y Montgomery, 96 B; x = y = 0 encodes infinity, which is not
// on the curve), accumulators are XYZZ (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; ZZ = 0 <=> infinity).
// Formulas: Explicit-Formulas Database, short Weierstrass a = 0, "xyzz" madd-2008-s (8M+2S),
// add-2008-s (12M+2S), dbl-2008-s-1, mdbl-2008-s-1 -- with every exceptional case handled
// (infinity operands, P = Q, P = -Q) so results are exact for any input, including repeated and
// infinite bases (SURVEY.md 7 "MSM exactness").
#pragma once
#include "ufield.cuh"

namespace zk {

using Fq = Fq381;
using FqE = Fe<Fq381>;

struct alignas(16) G1Affine {
    FqE x, y;
};
struct alignas(16) G1Xyzz {
    FqE x, y, zz, zzz;
};

ZK_HD bool g1_is_inf(const G1Affine &p) { return fe_is_zero<Fq>(p.x) && fe_is_zero<Fq>(p.y); }
ZK_HD bool g1_is_inf(const G1Xyzz &p) { return fe_is_zero<Fq>(p.zz); }

ZK_HD G1Xyzz g1_xyzz_inf() {
    G1Xyzz r;
    r.x = fe_zero<Fq>(); r.y = fe_zero<Fq>(); r.zz = fe_zero<Fq>(); r.zzz = fe_zero<Fq>();
    return r;
}
ZK_HD G1Xyzz g1_from_affine(const G1Affine &p) {
    G1Xyzz r;
    if (g1_is_inf(p)) return g1_xyzz_inf();
    r.x = p.x; r.y = p.y; r.zz = fe_one<Fq>(); r.zzz = fe_one<Fq>();
    return r;
}
ZK_HD G1Affine g1_neg(const G1Affine &p) {
    G1Affine r = p;
    r.y = fe_neg<Fq>(p.y);      // -0 = 0 keeps the infinity encoding
    return r;
}

// 2 * (affine p), p finite
ZK_HD G1Xyzz g1_mdbl(const G1Affine &p) {
    G1Xyzz r;
    FqE u = fe_dbl<Fq>(p.y);
    FqE v = fe_sqr<Fq>(u);
    FqE w = fe_mul<Fq>(u, v);
    FqE s = fe_mul<Fq>(p.x, v);
    FqE xx = fe_sqr<Fq>(p.x);
    FqE m = fe_add<Fq>(fe_dbl<Fq>(xx), xx);          // 3 X^2  (a = 0)
    r.x = fe_sub<Fq>(fe_sqr<Fq>(m), fe_dbl<Fq>(s));
    r.y = fe_sub<Fq>(fe_mul<Fq>(m, fe_sub<Fq>(s, r.x)), fe_mul<Fq>(w, p.y));
    r.zz = v;
    r.zzz = w;
    return r;
}

ZK_HD G1Xyzz g1_dbl(const G1Xyzz &p) {
    if (g1_is_inf(p)) return p;
    G1Xyzz r;
    FqE u = fe_dbl<Fq>(p.y);
    FqE v = fe_sqr<Fq>(u);
    FqE w = fe_mul<Fq>(u, v);
    FqE s = fe_mul<Fq>(p.x, v);
    FqE xx = fe_sqr<Fq>(p.x);
    FqE m = fe_add<Fq>(fe_dbl<Fq>(xx), xx);
    r.x = fe_sub<Fq>(fe_sqr<Fq>(m), fe_dbl<Fq>(s));
    r.y = fe_sub<Fq>(fe_mul<Fq>(m, fe_sub<Fq>(s, r.x)), fe_mul<Fq>(w, p.y));
    r.zz = fe_mul<Fq>(v, p.zz);
    r.zzz = fe_mul<Fq>(w, p.zzz);
    return r;
}

// acc + (affine q): the Pippenger bucket update
ZK_HD G1Xyzz g1_madd(const G1Xyzz &acc, const G1Affine &q) {
    if (g1_is_inf(q)) return acc;
    if (g1_is_inf(acc)) return g1_from_affine(q);
    FqE u2 = fe_mul<Fq>(q.x, acc.zz);
    FqE s2 = fe_mul<Fq>(q.y, acc.zzz);
    FqE p = fe_sub<Fq>(u2, acc.x);
    FqE r = fe_sub<Fq>(s2, acc.y);
    if (fe_is_zero<Fq>(p)) {
        if (fe_is_zero<Fq>(r)) return g1_mdbl(q);      // acc == q
        return g1_xyzz_inf();                          // acc == -q
    }
    G1Xyzz o;
    FqE pp = fe_sqr<Fq>(p);
    FqE ppp = fe_mul<Fq>(p, pp);
    FqE qq = fe_mul<Fq>(acc.x, pp);
    o.x = fe_sub<Fq>(fe_sub<Fq>(fe_sqr<Fq>(r), ppp), fe_dbl<Fq>(qq));
    o.y = fe_sub<Fq>(fe_mul<Fq>(r, fe_sub<Fq>(qq, o.x)), fe_mul<Fq>(acc.y, ppp));
    o.zz = fe_mul<Fq>(acc.zz, pp);
    o.zzz = fe_mul<Fq>(acc.zzz, ppp);
    return o;
}

ZK_HD G1Xyzz g1_add(const G1Xyzz &a, const G1Xyzz &b) {
    if (g1_is_inf(a)) return b;
    if (g1_is_inf(b)) return a;
    FqE u1 = fe_mul<Fq>(a.x, b.zz);
    FqE u2 = fe_mul<Fq>(b.x, a.zz);
    FqE s1 = fe_mul<Fq>(a.y, b.zzz);
    FqE s2 = fe_mul<Fq>(b.y, a.zzz);
    FqE p = fe_sub<Fq>(u2, u1);
    FqE r = fe_sub<Fq>(s2, s1);
    if (fe_is_zero<Fq>(p)) {
        if (fe_is_zero<Fq>(r)) return g1_dbl(a);
        return g1_xyzz_inf();
    }
    G1Xyzz o;
    FqE pp = fe_sqr<Fq>(p);
    FqE ppp = fe_mul<Fq>(p, pp);
    FqE qq = fe_mul<Fq>(u1, pp);
    o.x = fe_sub<Fq>(fe_sub<Fq>(fe_sqr<Fq>(r), ppp), fe_dbl<Fq>(qq));
    o.y = fe_sub<Fq>(fe_mul<Fq>(r, fe_sub<Fq>(qq, o.x)), fe_mul<Fq>(s1, ppp));
    o.zz = fe_mul<Fq>(fe_mul<Fq>(a.zz, b.zz), pp);
    o.zzz = fe_mul<Fq>(fe_mul<Fq>(a.zzz, b.zzz), ppp);
    return o;
}

// host-side normalisation (one inversion): x = X / ZZ, y = Y / ZZZ
inline G1Affine g1_to_affine(const G1Xyzz &p) {
    G1Affine r;
    if (g1_is_inf(p)) { r.x = fe_zero<Fq>(); r.y = fe_zero<Fq>(); return r; }
    // 1/ZZZ, then 1/ZZ = ZZZ^2/ZZ^3 * ... : use two products of a single inverse of ZZ*ZZZ
    FqE t = fe_inv<Fq>(fe_mul<Fq>(p.zz, p.zzz));
    FqE izz = fe_mul<Fq>(t, p.zzz), izzz = fe_mul<Fq>(t, p.zz);
    r.x = fe_mul<Fq>(p.x, izz);
    r.y = fe_mul<Fq>(p.y, izzz);
    return r;
}

ZK_HD bool g1_on_curve(const G1Affine &p) {
    if (g1_is_inf(p)) return true;
    FqE lhs = fe_sqr<Fq>(p.y);
    FqE rhs = fe_add<Fq>(fe_mul<Fq>(fe_sqr<Fq>(p.x), p.x), fe_from_u64<Fq>(4));
    return fe_eq<Fq>(lhs, rhs);
}

// generator (canonical coordinates), returned in Montgomery form
inline G1Affine g1_generator() {
    FqE cx, cy;
    const uint32_t gx[12] = {0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu,
                             0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u};
    const uint32_t gy[12] = {0x46c5e7e1u, 0x0caa2329u, 0xa2888ae4u, 0xd03cc744u, 0x2c04b3edu, 0x00db18cbu,
                             0xd5d00af6u, 0xfcf5e095u, 0x741d8ae4u, 0xa09e30edu, 0xe3aaa0f1u, 0x08b3f481u};
    for (int i = 0; i < 12; i++) { cx.l[i] = gx[i]; cy.l[i] = gy[i]; }
    G1Affine g;
    g.x = fe_from_canonical<Fq>(cx);
    g.y = fe_from_canonical<Fq>(cy);
    return g;
}

// [k] p for a canonical little-endian scalar of `nlimbs` 32-bit limbs (double-and-add, MSB first)
ZK_HD G1Xyzz g1_mul_canonical(const G1Affine &p, const uint32_t *k, int nlimbs) {
    G1Xyzz acc = g1_xyzz_inf();
    for (int i = 32 * nlimbs - 1; i >= 0; i--) {
        acc = g1_dbl(acc);
        if ((k[i / 32] >> (i % 32)) & 1) acc = g1_madd(acc, p);
    }
    return acc;
}

// 16-byte vector loads / stores of points
__device__ __forceinline__ G1Affine g1_load_affine(const void *base, size_t idx) {
    G1Affine p;
    p.x = fe_load<Fq>(base, 2 * idx);
    p.y = fe_load<Fq>(base, 2 * idx + 1);
    return p;
}
__device__ __forceinline__ void g1_store_affine(void *base, size_t idx, const G1Affine &p) {
    fe_store<Fq>(base, 2 * idx, p.x);
    fe_store<Fq>(base, 2 * idx + 1, p.y);
}
__device__ __forceinline__ G1Xyzz g1_load_xyzz(const void *base, size_t idx) {
    G1Xyzz p;
    p.x = fe_load<Fq>(base, 4 * idx); p.y = fe_load<Fq>(base, 4 * idx + 1);
    p.zz = fe_load<Fq>(base, 4 * idx + 2); p.zzz = fe_load<Fq>(base, 4 * idx + 3);
    return p;
}
__device__ __forceinline__ void g1_store_xyzz(void *base, size_t idx, const G1Xyzz &p) {
    fe_store<Fq>(base, 4 * idx, p.x); fe_store<Fq>(base, 4 * idx + 1, p.y);
    fe_store<Fq>(base, 4 * idx + 2, p.zz); fe_store<Fq>(base, 4 * idx + 3, p.zzz);
}

}  // namespace zk
