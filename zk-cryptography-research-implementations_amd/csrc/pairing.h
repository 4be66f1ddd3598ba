// pairing.h -- host-side BLS12-381 extension tower, G2 and the optimal ate pairing: the verifier's control path.
//
// MultilinearKZG::verify (multilinear_kzg/src/multilinear_kzg.rs:131-158) checks
//     e(C - [v] G1, G2) == prod_i e(pi_i, [tau_i] G2 - [x_i] G2)
// with n + 1 pairings of ark-bls12-381 0.5.0 ([ext], not vendored).  This is O(n) work on a handful of points, a separate
// arithmetic tower with no table-sized loop, so it runs on the host like the transcript; the GPU path ends at
// the commitment and the proofs.  Restated from the published curve definition:
//   Fq2 = Fq[u]/(u^2 + 1), Fq6 = Fq2[v]/(v^3 - xi), xi = 1 + u, Fq12 = Fq6[w]/(w^2 - v); E': y^2 = x^3 + 4 xi (M-type twist);
//   x = -0xd201000000010000; e(P, Q) = conj(f_{|x|,Q}(P))^((p^12 - 1)/r).
// Design: G2 stays on the twist (Fq2, affine; the n + 1 slope denominators of a step share ONE inversion), lines are
// evaluated in the sparse form  l w^3 = (lambda x_T - y_T) - lambda x_P v + y_P v w  (w^3 lies in Fq4, so the factor dies in
// the final exponentiation), all pairings share one Miller accumulator, and the final exponentiation is the easy part
// (p^6 - 1)(p^2 + 1) followed by one power by (p^4 - p^2 + 1)/r.  oracle/pairing_model.py does all of this the slow obvious way
// (untwisted points, generic Fq12 lines, one power by (p^12 - 1)/r) and must produce the same element of GT.
#pragma once
#include <vector>

#include "g1.cuh"
#include "pairing_consts.h"

namespace zk {
namespace pairing {

inline FqE fq_from_limbs64(const uint64_t *c) {
    FqE t;
    for (int i = 0; i < 6; i++) { t.l[2 * i] = (uint32_t)c[i]; t.l[2 * i + 1] = (uint32_t)(c[i] >> 32); }
    return fe_from_canonical<Fq>(t);
}

// ---- Fq2 ---------------------------------------------------------------------------------------------------------------------
struct Fq2 {
    FqE c0, c1;
};
inline Fq2 f2_zero() { return Fq2{fe_zero<Fq>(), fe_zero<Fq>()}; }
inline Fq2 f2_one() { return Fq2{fe_one<Fq>(), fe_zero<Fq>()}; }
inline Fq2 f2_from_fq(const FqE &a) { return Fq2{a, fe_zero<Fq>()}; }
inline bool f2_is_zero(const Fq2 &a) { return fe_is_zero<Fq>(a.c0) && fe_is_zero<Fq>(a.c1); }
inline bool f2_eq(const Fq2 &a, const Fq2 &b) { return fe_eq<Fq>(a.c0, b.c0) && fe_eq<Fq>(a.c1, b.c1); }
inline Fq2 f2_add(const Fq2 &a, const Fq2 &b) { return Fq2{fe_add<Fq>(a.c0, b.c0), fe_add<Fq>(a.c1, b.c1)}; }
inline Fq2 f2_sub(const Fq2 &a, const Fq2 &b) { return Fq2{fe_sub<Fq>(a.c0, b.c0), fe_sub<Fq>(a.c1, b.c1)}; }
inline Fq2 f2_neg(const Fq2 &a) { return Fq2{fe_neg<Fq>(a.c0), fe_neg<Fq>(a.c1)}; }
inline Fq2 f2_dbl(const Fq2 &a) { return f2_add(a, a); }
inline Fq2 f2_mul(const Fq2 &a, const Fq2 &b) {             // 3 products (Karatsuba), u^2 = -1
    FqE t0 = fe_mul<Fq>(a.c0, b.c0), t1 = fe_mul<Fq>(a.c1, b.c1);
    FqE s = fe_mul<Fq>(fe_add<Fq>(a.c0, a.c1), fe_add<Fq>(b.c0, b.c1));
    return Fq2{fe_sub<Fq>(t0, t1), fe_sub<Fq>(fe_sub<Fq>(s, t0), t1)};
}
inline Fq2 f2_sqr(const Fq2 &a) {
    FqE t = fe_mul<Fq>(a.c0, a.c1);
    return Fq2{fe_mul<Fq>(fe_add<Fq>(a.c0, a.c1), fe_sub<Fq>(a.c0, a.c1)), fe_dbl<Fq>(t)};
}
inline Fq2 f2_mul_fq(const Fq2 &a, const FqE &k) { return Fq2{fe_mul<Fq>(a.c0, k), fe_mul<Fq>(a.c1, k)}; }
inline Fq2 f2_mul_xi(const Fq2 &a) { return Fq2{fe_sub<Fq>(a.c0, a.c1), fe_add<Fq>(a.c0, a.c1)}; }   // (1 + u) a
inline Fq2 f2_inv(const Fq2 &a) {
    FqE d = fe_inv<Fq>(fe_add<Fq>(fe_sqr<Fq>(a.c0), fe_sqr<Fq>(a.c1)));
    return Fq2{fe_mul<Fq>(a.c0, d), fe_neg<Fq>(fe_mul<Fq>(a.c1, d))};
}
// all inverses with one field inversion (Montgomery's trick); zeros are left as zeros
inline void f2_batch_inv(std::vector<Fq2> &v) {
    std::vector<Fq2> prefix(v.size());
    Fq2 run = f2_one();
    for (size_t i = 0; i < v.size(); i++) {
        prefix[i] = run;
        if (!f2_is_zero(v[i])) run = f2_mul(run, v[i]);
    }
    Fq2 inv = f2_inv(run);
    for (size_t i = v.size(); i-- > 0;) {
        if (f2_is_zero(v[i])) continue;
        Fq2 t = f2_mul(inv, prefix[i]);
        inv = f2_mul(inv, v[i]);
        v[i] = t;
    }
}

// ---- Fq6 = Fq2[v] / (v^3 - xi) ----------------------------------------------------------------------------------------------------
struct Fq6 {
    Fq2 c0, c1, c2;
};
inline Fq6 f6_zero() { return Fq6{f2_zero(), f2_zero(), f2_zero()}; }
inline Fq6 f6_one() { return Fq6{f2_one(), f2_zero(), f2_zero()}; }
inline Fq6 f6_add(const Fq6 &a, const Fq6 &b) { return Fq6{f2_add(a.c0, b.c0), f2_add(a.c1, b.c1), f2_add(a.c2, b.c2)}; }
inline Fq6 f6_sub(const Fq6 &a, const Fq6 &b) { return Fq6{f2_sub(a.c0, b.c0), f2_sub(a.c1, b.c1), f2_sub(a.c2, b.c2)}; }
inline Fq6 f6_neg(const Fq6 &a) { return Fq6{f2_neg(a.c0), f2_neg(a.c1), f2_neg(a.c2)}; }
inline bool f6_eq(const Fq6 &a, const Fq6 &b) { return f2_eq(a.c0, b.c0) && f2_eq(a.c1, b.c1) && f2_eq(a.c2, b.c2); }
inline Fq6 f6_mul(const Fq6 &a, const Fq6 &b) {             // 6 Fq2 products
    Fq2 t0 = f2_mul(a.c0, b.c0), t1 = f2_mul(a.c1, b.c1), t2 = f2_mul(a.c2, b.c2);
    Fq2 c0 = f2_add(t0, f2_mul_xi(f2_sub(f2_sub(f2_mul(f2_add(a.c1, a.c2), f2_add(b.c1, b.c2)), t1), t2)));
    Fq2 c1 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a.c0, a.c1), f2_add(b.c0, b.c1)), t0), t1), f2_mul_xi(t2));
    Fq2 c2 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a.c0, a.c2), f2_add(b.c0, b.c2)), t0), t2), t1);
    return Fq6{c0, c1, c2};
}
inline Fq6 f6_mul_by_v(const Fq6 &a) { return Fq6{f2_mul_xi(a.c2), a.c0, a.c1}; }
inline Fq6 f6_inv(const Fq6 &a) {
    Fq2 A = f2_sub(f2_sqr(a.c0), f2_mul_xi(f2_mul(a.c1, a.c2)));
    Fq2 B = f2_sub(f2_mul_xi(f2_sqr(a.c2)), f2_mul(a.c0, a.c1));
    Fq2 Cc = f2_sub(f2_sqr(a.c1), f2_mul(a.c0, a.c2));
    Fq2 F = f2_add(f2_mul(a.c0, A), f2_mul_xi(f2_add(f2_mul(a.c2, B), f2_mul(a.c1, Cc))));
    Fq2 Fi = f2_inv(F);
    return Fq6{f2_mul(A, Fi), f2_mul(B, Fi), f2_mul(Cc, Fi)};
}

// ---- Fq12 = Fq6[w] / (w^2 - v) ----------------------------------------------------------------------------------------------------
struct Fq12 {
    Fq6 c0, c1;
};
inline Fq12 f12_one() { return Fq12{f6_one(), f6_zero()}; }
inline bool f12_eq(const Fq12 &a, const Fq12 &b) { return f6_eq(a.c0, b.c0) && f6_eq(a.c1, b.c1); }
inline Fq12 f12_mul(const Fq12 &a, const Fq12 &b) {         // 3 Fq6 products
    Fq6 t0 = f6_mul(a.c0, b.c0), t1 = f6_mul(a.c1, b.c1);
    Fq6 c1 = f6_sub(f6_sub(f6_mul(f6_add(a.c0, a.c1), f6_add(b.c0, b.c1)), t0), t1);
    return Fq12{f6_add(t0, f6_mul_by_v(t1)), c1};
}
inline Fq12 f12_sqr(const Fq12 &a) {                        // (c0 + c1 w)^2 with 2 Fq6 products
    Fq6 ab = f6_mul(a.c0, a.c1);
    Fq6 t = f6_mul(f6_add(a.c0, a.c1), f6_add(a.c0, f6_mul_by_v(a.c1)));
    return Fq12{f6_sub(f6_sub(t, ab), f6_mul_by_v(ab)), f6_add(ab, ab)};
}
inline Fq12 f12_conj(const Fq12 &a) { return Fq12{a.c0, f6_neg(a.c1)}; }     // w -> -w, the p^6 Frobenius
inline Fq12 f12_inv(const Fq12 &a) {
    Fq6 n = f6_inv(f6_sub(f6_mul(a.c0, a.c0), f6_mul_by_v(f6_mul(a.c1, a.c1))));
    return Fq12{f6_mul(a.c0, n), f6_neg(f6_mul(a.c1, n))};
}
// coefficient of w^k, k = 0..5 (w^2 = v): c0.c0, c1.c0, c0.c1, c1.c1, c0.c2, c1.c2
inline Fq2 &f12_coeff(Fq12 &a, int k) {
    Fq6 &h = (k & 1) ? a.c1 : a.c0;
    return (k >> 1) == 0 ? h.c0 : ((k >> 1) == 1 ? h.c1 : h.c2);
}
// x -> x^(p^2): Fq2 coefficients are fixed, w^(p^2) = gamma w with gamma = xi^((p^2-1)/6) in Fq
inline Fq12 f12_frobenius_p2(const Fq12 &a) {
    static const FqE gamma = fq_from_limbs64(kGamma);
    Fq12 r = a;
    FqE g = fe_one<Fq>();
    for (int k = 1; k < 6; k++) {
        g = fe_mul<Fq>(g, gamma);
        Fq2 &c = f12_coeff(r, k);
        c = f2_mul_fq(c, g);
    }
    return r;
}
// the sparse line value (A, B, C) = A + B v + C v w
inline Fq12 f12_mul_line(const Fq12 &f, const Fq2 &A, const Fq2 &B, const Fq2 &Cc) {
    Fq12 l{Fq6{A, B, f2_zero()}, Fq6{f2_zero(), Cc, f2_zero()}};
    return f12_mul(f, l);
}

// ---- G2 on the twist E'(Fq2): y^2 = x^3 + 4 xi; affine, (0, 0) = infinity --------------------------------------------------------------
struct G2Affine {
    Fq2 x, y;
};
inline bool g2_is_inf(const G2Affine &p) { return f2_is_zero(p.x) && f2_is_zero(p.y); }
inline G2Affine g2_inf() { return G2Affine{f2_zero(), f2_zero()}; }
inline G2Affine g2_neg(const G2Affine &p) { return G2Affine{p.x, f2_neg(p.y)}; }
inline G2Affine g2_generator() {
    return G2Affine{Fq2{fq_from_limbs64(kG2x0), fq_from_limbs64(kG2x1)}, Fq2{fq_from_limbs64(kG2y0), fq_from_limbs64(kG2y1)}};
}
inline bool g2_on_curve(const G2Affine &p) {
    if (g2_is_inf(p)) return true;
    Fq2 b = f2_mul_xi(f2_from_fq(fe_from_u64<Fq>(4)));
    return f2_eq(f2_sqr(p.y), f2_add(f2_mul(f2_sqr(p.x), p.x), b));
}
struct G2Jac {                                              // Jacobian: x = X / Z^2, y = Y / Z^3; Z = 0 is infinity
    Fq2 x, y, z;
};
inline G2Jac g2j_inf() { return G2Jac{f2_one(), f2_one(), f2_zero()}; }
inline G2Jac g2j_from_affine(const G2Affine &p) { return g2_is_inf(p) ? g2j_inf() : G2Jac{p.x, p.y, f2_one()}; }
inline G2Jac g2j_dbl(const G2Jac &p) {                      // a = 0
    if (f2_is_zero(p.z)) return p;
    Fq2 A = f2_sqr(p.x), B = f2_sqr(p.y), Cc = f2_sqr(B);
    Fq2 D = f2_dbl(f2_sub(f2_sub(f2_sqr(f2_add(p.x, B)), A), Cc));
    Fq2 E = f2_add(f2_dbl(A), A), Fv = f2_sqr(E);
    G2Jac r;
    r.x = f2_sub(Fv, f2_dbl(D));
    r.y = f2_sub(f2_mul(E, f2_sub(D, r.x)), f2_dbl(f2_dbl(f2_dbl(Cc))));
    r.z = f2_dbl(f2_mul(p.y, p.z));
    return r;
}
inline G2Jac g2j_add(const G2Jac &a, const G2Jac &b) {
    if (f2_is_zero(a.z)) return b;
    if (f2_is_zero(b.z)) return a;
    Fq2 z1z1 = f2_sqr(a.z), z2z2 = f2_sqr(b.z);
    Fq2 u1 = f2_mul(a.x, z2z2), u2 = f2_mul(b.x, z1z1);
    Fq2 s1 = f2_mul(f2_mul(a.y, b.z), z2z2), s2 = f2_mul(f2_mul(b.y, a.z), z1z1);
    Fq2 h = f2_sub(u2, u1), rr = f2_sub(s2, s1);
    if (f2_is_zero(h)) return f2_is_zero(rr) ? g2j_dbl(a) : g2j_inf();
    Fq2 hh = f2_sqr(h), hhh = f2_mul(h, hh), v = f2_mul(u1, hh);
    G2Jac r;
    r.x = f2_sub(f2_sub(f2_sqr(rr), hhh), f2_dbl(v));
    r.y = f2_sub(f2_mul(rr, f2_sub(v, r.x)), f2_mul(s1, hhh));
    r.z = f2_mul(f2_mul(a.z, b.z), h);
    return r;
}
inline G2Affine g2j_to_affine(const G2Jac &p) {
    if (f2_is_zero(p.z)) return g2_inf();
    Fq2 zi = f2_inv(p.z), zi2 = f2_sqr(zi);
    return G2Affine{f2_mul(p.x, zi2), f2_mul(p.y, f2_mul(zi2, zi))};
}
// [k] p, k = canonical little-endian 32-bit limbs (mul_bigint: double-and-add, MSB first)
inline G2Jac g2_mul_canonical(const G2Affine &p, const uint32_t *k, int nlimbs) {
    G2Jac acc = g2j_inf(), base = g2j_from_affine(p);
    for (int i = 32 * nlimbs - 1; i >= 0; i--) {
        acc = g2j_dbl(acc);
        if ((k[i / 32] >> (i % 32)) & 1) acc = g2j_add(acc, base);
    }
    return acc;
}

// ---- the pairing product ---------------------------------------------------------------------------------------------------------
struct PairIn {
    G1Affine p;
    G2Affine q;
};
// prod_i f_{|x|, Q_i}(P_i), conjugated (x < 0); pairs with an infinite member contribute 1
inline Fq12 multi_miller_loop(const std::vector<PairIn> &in) {
    struct St { FqE xp, yp; Fq2 xq, yq, xt, yt; };
    std::vector<St> s;
    for (const PairIn &pr : in)
        if (!g1_is_inf(pr.p) && !g2_is_inf(pr.q)) s.push_back(St{pr.p.x, pr.p.y, pr.q.x, pr.q.y, pr.q.x, pr.q.y});
    Fq12 f = f12_one();
    std::vector<Fq2> den(s.size());
    for (int bit = 62; bit >= 0; bit--) {                   // kXAbs has its top bit at position 63
        f = f12_sqr(f);
        for (size_t i = 0; i < s.size(); i++) den[i] = f2_dbl(s[i].yt);
        f2_batch_inv(den);
        for (size_t i = 0; i < s.size(); i++) {             // tangent at T, then T = 2 T
            St &t = s[i];
            Fq2 xx = f2_sqr(t.xt);
            Fq2 lam = f2_mul(f2_add(f2_dbl(xx), xx), den[i]);
            f = f12_mul_line(f, f2_sub(f2_mul(lam, t.xt), t.yt), f2_neg(f2_mul_fq(lam, t.xp)), f2_from_fq(t.yp));
            Fq2 x3 = f2_sub(f2_sqr(lam), f2_dbl(t.xt));
            t.yt = f2_sub(f2_mul(lam, f2_sub(t.xt, x3)), t.yt);
            t.xt = x3;
        }
        if ((kXAbs >> bit) & 1) {
            for (size_t i = 0; i < s.size(); i++) den[i] = f2_sub(s[i].xq, s[i].xt);
            f2_batch_inv(den);
            for (size_t i = 0; i < s.size(); i++) {         // chord through T and Q, then T = T + Q
                St &t = s[i];
                Fq2 lam = f2_mul(f2_sub(t.yq, t.yt), den[i]);
                f = f12_mul_line(f, f2_sub(f2_mul(lam, t.xt), t.yt), f2_neg(f2_mul_fq(lam, t.xp)), f2_from_fq(t.yp));
                Fq2 x3 = f2_sub(f2_sub(f2_sqr(lam), t.xt), t.xq);
                t.yt = f2_sub(f2_mul(lam, f2_sub(t.xt, x3)), t.yt);
                t.xt = x3;
            }
        }
    }
    return f12_conj(f);
}
inline Fq12 final_exponentiation(const Fq12 &f) {
    Fq12 t = f12_mul(f12_conj(f), f12_inv(f));              // f^(p^6 - 1)
    t = f12_mul(f12_frobenius_p2(t), t);                    // ^(p^2 + 1)
    Fq12 acc = f12_one();                                   // ^((p^4 - p^2 + 1) / r), MSB first
    bool started = false;
    for (int i = 64 * kHardLimbs - 1; i >= 0; i--) {
        if (started) acc = f12_sqr(acc);
        if ((kHardExp[i / 64] >> (i % 64)) & 1) {
            acc = started ? f12_mul(acc, t) : t;
            started = true;
        }
    }
    return acc;
}
inline Fq12 pairing_product(const std::vector<PairIn> &in) { return final_exponentiation(multi_miller_loop(in)); }

}  // namespace pairing
}  // namespace zk
