// g1u.cuh -- the Pippenger bucket update in the unsaturated (14 x 29-bit limb) Fq representation.
//
// Same group law as g1.cuh's madd-2008-s, rearranged so that every subtrahend is a product (< 2 p)
// or a doubled product (< 4 p): with P' = X1 - U2 and R' = Y1 - S2 (the negated differences)
//     PP = P'^2, PPP' = P' PP, Q = X1 PP,
//     X3 = R'^2 + PPP' - 2 Q,   Y3 = R' (X3 - Q) + Y1 PPP',   ZZ3 = ZZ1 PP,   ZZZ3 = -(ZZZ1 PPP').
// No value ever needs a modular reduction (inputs of a product may be any representative below
// 2^12 p; accumulator coordinates stay below 8 p), additions are plain limb adds with one carry
// pass, and the exceptional cases (P = Q, P = -Q) are detected AFTER the fact from ZZ3 = 0 mod p
// and redone exactly on a rare slow path.  Bases are stored pre-converted (2 x 16 words = one
// 128-byte line per point); results leave the kernel in the stored XYZZ form of g1.cuh.
#pragma once
#include "g1.cuh"

namespace zk {

using FqU = Ufe<Fq381>;
constexpr int kUWords = 16;                         // 14 limbs + 2 pad words: 64 B per coordinate

struct G1AffineU { FqU x, y; };
struct G1XyzzU { FqU x, y, zz, zzz; bool inf; };

ZK_HD FqU fqu_one() {                               // the internal Montgomery one: 2^(29 L) mod p
    FqU r;
#pragma unroll
    for (int j = 0; j < UParams<Fq381>::L; j++) r.l[j] = UParams<Fq381>::r_u(j);
    return r;
}
ZK_HD FqU fqu_renorm(const FqU &a) { return umul<Fq381>(a, fqu_one()); }      // any representative -> (< 2 p)
ZK_HD bool fqu_is_zero(const FqU &a) { return u_is_zero_mod_p<Fq381>(fqu_renorm(a)); }

__device__ __forceinline__ FqU fqu_load(const uint32_t *p) {
    FqU r;
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w; r.l[12] = d.x; r.l[13] = d.y;
    return r;
}
__device__ __forceinline__ void fqu_store(uint32_t *p, const FqU &v) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    q[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
    q[3] = make_uint4(v.l[12], v.l[13], 0u, 0u);
}
__device__ __forceinline__ G1AffineU g1u_load_affine(const void *base, size_t idx) {
    const uint32_t *p = reinterpret_cast<const uint32_t *>(base) + idx * (2 * kUWords);
    G1AffineU r;
    r.x = fqu_load(p);
    r.y = fqu_load(p + kUWords);
    return r;
}

// exact doubling of an affine point, all in the internal form (rare path)
__device__ __forceinline__ void g1u_mdbl(G1XyzzU &o, const G1AffineU &q) {
    using F = Fq381;
    FqU u = uadd<F>(q.y, q.y);
    FqU v = usqr<F>(u);
    FqU w = umul<F>(u, v);
    FqU s = umul<F>(q.x, v);
    FqU xx = usqr<F>(q.x);
    FqU m = uadd<F>(uadd<F>(xx, xx), xx);
    FqU x3 = fqu_renorm(usub<F>(usqr<F>(m), uadd<F>(s, s)));      // < 2 p again
    o.x = x3;
    o.y = usub<F>(umul<F>(m, usub<F>(s, x3)), umul<F>(w, q.y));
    o.zz = v;
    o.zzz = w;
    o.inf = false;
}

// acc += q  (q finite or the exact-zero infinity encoding; `neg` adds -q)
__device__ __forceinline__ void g1u_madd(G1XyzzU &acc, G1AffineU q, bool neg) {
    using F = Fq381;
    if (u_is_exact_zero<F>(q.x) && u_is_exact_zero<F>(q.y)) return;           // base at infinity
    if (neg) q.y = usub<F>(u_zero<F>(), q.y);                                  // 4 p - y
    if (acc.inf) {
        acc.x = q.x; acc.y = q.y; acc.zz = fqu_one(); acc.zzz = fqu_one(); acc.inf = false;
        return;
    }
    FqU u2 = umul<F>(q.x, acc.zz);
    FqU s2 = umul<F>(q.y, acc.zzz);
    FqU pn = usub<F>(acc.x, u2);                       // P' = X1 - U2
    FqU rn = usub<F>(acc.y, s2);                       // R' = Y1 - S2
    FqU pp = usqr<F>(pn);
    FqU pppn = umul<F>(pn, pp);                        // PPP' = -PPP
    FqU qq = umul<F>(acc.x, pp);
    FqU zz3 = umul<F>(acc.zz, pp);
    if (u_is_zero_mod_p<F>(zz3)) {                     // P = 0 (or ZZ1 = 0 mod p): exceptional, redo exactly
        if (fqu_is_zero(pn)) {
            if (fqu_is_zero(rn)) g1u_mdbl(acc, q);     // acc == q
            else { acc.inf = true; acc.x = u_zero<F>(); acc.y = u_zero<F>(); acc.zz = u_zero<F>(); acc.zzz = u_zero<F>(); }
            return;
        }
    }
    FqU x3 = usub<F>(uadd<F>(usqr<F>(rn), pppn), uadd<F>(qq, qq));
    FqU y3 = umul2<F>(rn, usub<F>(x3, qq), acc.y, pppn);      // one reduction for the two products
    acc.zzz = usub<F>(u_zero<F>(), umul<F>(acc.zzz, pppn));
    acc.zz = zz3;
    acc.x = x3;
    acc.y = y3;
}

// ---- XYZZ points kept in the internal form between kernels (partial sums, the bucket reduction arrays) ---------------------------
// 4 x 64 B per point; infinity is the all-zero record (ZZ exactly 0), as in the stored form.
static_assert(kXyzzUBytes == 4 * kUWords * 4, "msm_kernels.cuh");
__device__ __forceinline__ G1XyzzU g1u_inf() {
    G1XyzzU r;
    r.x = u_zero<Fq381>(); r.y = u_zero<Fq381>(); r.zz = u_zero<Fq381>(); r.zzz = u_zero<Fq381>();
    r.inf = true;
    return r;
}
__device__ __forceinline__ G1XyzzU g1u_load_xyzz(const void *base, size_t idx) {
    const uint32_t *p = reinterpret_cast<const uint32_t *>(base) + idx * (4 * kUWords);
    G1XyzzU r;
    r.x = fqu_load(p);
    r.y = fqu_load(p + kUWords);
    r.zz = fqu_load(p + 2 * kUWords);
    r.zzz = fqu_load(p + 3 * kUWords);
    r.inf = u_is_exact_zero<Fq381>(r.zz);
    return r;
}
__device__ __forceinline__ void g1u_store_xyzz(void *base, size_t idx, const G1XyzzU &a) {
    uint32_t *p = reinterpret_cast<uint32_t *>(base) + idx * (4 * kUWords);
    const G1XyzzU v = a.inf ? g1u_inf() : a;
    fqu_store(p, v.x);
    fqu_store(p + kUWords, v.y);
    fqu_store(p + 2 * kUWords, v.zz);
    fqu_store(p + 3 * kUWords, v.zzz);
}
// 2 a  (dbl-2008-s-1 on y^2 = x^3 + b).  Coordinates of `a` below 8 p; the result's below 6 p.
__device__ __forceinline__ G1XyzzU g1u_dbl(const G1XyzzU &a) {
    using F = Fq381;
    if (a.inf) return a;
    G1XyzzU o;
    FqU u = uadd<F>(a.y, a.y);
    FqU v = usqr<F>(u);
    FqU w = umul<F>(u, v);
    FqU s = umul<F>(a.x, v);
    FqU xx = usqr<F>(a.x);
    FqU m = uadd<F>(uadd<F>(xx, xx), xx);
    FqU x3 = fqu_renorm(usub<F>(usqr<F>(m), uadd<F>(s, s)));      // < 2 p, so that it can be a subtrahend
    o.x = x3;
    o.y = usub<F>(umul<F>(m, usub<F>(s, x3)), umul<F>(w, a.y));
    o.zz = umul<F>(v, a.zz);
    o.zzz = umul<F>(w, a.zzz);
    o.inf = false;
    return o;
}
// a + b, both XYZZ (add-2008-s with the negated differences of g1u_madd; 12 products + 2 squarings, Y3 as one dual product).
// Coordinates of the operands below 8 p; the result's below 8 p.  P = Q and P = -Q are detected from ZZ3 = 0 mod p.
__device__ __forceinline__ G1XyzzU g1u_add(const G1XyzzU &a, const G1XyzzU &b) {
    using F = Fq381;
    if (a.inf) return b;
    if (b.inf) return a;
    FqU u1 = umul<F>(a.x, b.zz), u2 = umul<F>(b.x, a.zz);
    FqU s1 = umul<F>(a.y, b.zzz), s2 = umul<F>(b.y, a.zzz);
    FqU pn = usub<F>(u1, u2);                          // P' = U1 - U2
    FqU rn = usub<F>(s1, s2);                          // R' = S1 - S2
    FqU pp = usqr<F>(pn);
    FqU pppn = umul<F>(pn, pp);
    FqU qq = umul<F>(u1, pp);
    FqU zz3 = umul<F>(umul<F>(a.zz, b.zz), pp);
    if (u_is_zero_mod_p<F>(zz3)) {                     // P' = 0: the same x coordinate
        if (fqu_is_zero(pn)) return fqu_is_zero(rn) ? g1u_dbl(a) : g1u_inf();
    }
    G1XyzzU o;
    FqU x3 = usub<F>(uadd<F>(usqr<F>(rn), pppn), uadd<F>(qq, qq));
    o.y = umul2<F>(rn, usub<F>(x3, qq), s1, pppn);
    o.zzz = usub<F>(u_zero<F>(), umul<F>(umul<F>(a.zzz, b.zzz), pppn));
    o.zz = zz3;
    o.x = x3;
    o.inf = false;
    return o;
}

// ---- one point operation by the four lanes of a quad ---------------------------------------------------------------------------
// The late levels of the bucket reduction have a handful of additions each: a level lasts as long as ONE lane's chain of 14
// products.  Here the four lanes of a DPP quad hold the same operands, each takes one of the (up to four) independent products
// of a stage, and the results are broadcast inside the quad with quad_perm moves (VALU rate, no LDS): 4 product-times per
// addition instead of 14, 3-4 per doubling instead of 10.  All four lanes return the same point.
template <int J> __device__ __forceinline__ FqU quad_bcast(const FqU &v) {
    FqU r;
#pragma unroll
    for (int i = 0; i < UParams<Fq381>::L; i++)
        r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], J * 0x55, 0xf, 0xf, false);
    return r;
}
__device__ __forceinline__ FqU quad_sel(unsigned q, const FqU &a0, const FqU &a1, const FqU &a2, const FqU &a3) {
    FqU r;
#pragma unroll
    for (int i = 0; i < UParams<Fq381>::L; i++) r.l[i] = q == 0 ? a0.l[i] : (q == 1 ? a1.l[i] : (q == 2 ? a2.l[i] : a3.l[i]));
    return r;
}
// The lanes of a wave run the quad operations in LOCKSTEP: no lane leaves early.  r4: with `if (a.inf) return a;` in front of the stages, a wave in which
// some quads held infinity and others did not produced wrong doublings in the quads that went on (tools/test_quad_ops.hip reproduces it: a branch in
// front of the quad_perm moves); r3's kernels only ever met all-infinity or no-infinity waves there.  So every lane computes every stage -- the all-zero
// record goes through the formulas as zeros -- and the cases are SELECTED at the end, limb by limb; the one real branch left (equal x: doubling /
// cancellation, redone exactly by every lane alone) comes after the last cross-lane move.
__device__ __forceinline__ FqU fqu_select(bool c, const FqU &a, const FqU &b) {
    FqU r;
#pragma unroll
    for (int i = 0; i < UParams<Fq381>::L; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}
__device__ __forceinline__ G1XyzzU g1u_select(bool c, const G1XyzzU &a, const G1XyzzU &b) {
    G1XyzzU r;
    r.x = fqu_select(c, a.x, b.x); r.y = fqu_select(c, a.y, b.y); r.zz = fqu_select(c, a.zz, b.zz); r.zzz = fqu_select(c, a.zzz, b.zzz);
    r.inf = c ? a.inf : b.inf;
    return r;
}
__device__ __forceinline__ G1XyzzU g1u_dbl_quad(const G1XyzzU &a, unsigned q) {
    using F = Fq381;
    const FqU u = uadd<F>(a.y, a.y);
    // stage 1: V = U^2 | XX = X^2
    FqU t = umul<F>(quad_sel(q, u, a.x, u, a.x), quad_sel(q, u, a.x, u, a.x));
    const FqU v = quad_bcast<0>(t), xx = quad_bcast<1>(t);
    const FqU m = uadd<F>(uadd<F>(xx, xx), xx);
    // stage 2: W = U V | S = X V | MM = M^2 | ZZ3 = V ZZ
    t = umul<F>(quad_sel(q, u, a.x, m, v), quad_sel(q, v, v, m, a.zz));
    const FqU w = quad_bcast<0>(t), s = quad_bcast<1>(t), mm = quad_bcast<2>(t), zz3 = quad_bcast<3>(t);
    const FqU x3u = usub<F>(mm, uadd<F>(s, s));
    // stage 3: X3 renormalised (< 2 p: it is a subtrahend) | W Y | W ZZZ
    t = umul<F>(quad_sel(q, x3u, w, w, w), quad_sel(q, fqu_one(), a.y, a.zzz, a.zzz));
    const FqU x3 = quad_bcast<0>(t), wy = quad_bcast<1>(t), zzz3 = quad_bcast<2>(t);
    // stage 4: M (S - X3)
    const FqU y3a = umul<F>(m, usub<F>(s, x3));
    G1XyzzU o;
    o.x = x3;
    o.y = usub<F>(y3a, wy);
    o.zz = zz3;
    o.zzz = zzz3;
    o.inf = false;
    return g1u_select(a.inf, a, o);
}
__device__ __forceinline__ G1XyzzU g1u_add_quad(const G1XyzzU &a, const G1XyzzU &b, unsigned q) {
    using F = Fq381;
    // stage 1: U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1
    FqU t = umul<F>(quad_sel(q, a.x, b.x, a.y, b.y), quad_sel(q, b.zz, a.zz, b.zzz, a.zzz));
    const FqU u1 = quad_bcast<0>(t), u2 = quad_bcast<1>(t), s1 = quad_bcast<2>(t), s2 = quad_bcast<3>(t);
    const FqU pn = usub<F>(u1, u2), rn = usub<F>(s1, s2);
    // stage 2: PP = P'^2 | RR = R'^2 | ZZ1 ZZ2 | ZZZ1 ZZZ2
    t = umul<F>(quad_sel(q, pn, rn, a.zz, a.zzz), quad_sel(q, pn, rn, b.zz, b.zzz));
    const FqU pp = quad_bcast<0>(t), rr = quad_bcast<1>(t), zz12 = quad_bcast<2>(t), zzz12 = quad_bcast<3>(t);
    // stage 3: PPP' = P' PP | Q = U1 PP | ZZ3 = ZZ12 PP
    t = umul<F>(quad_sel(q, pn, u1, zz12, zz12), pp);
    const FqU pppn = quad_bcast<0>(t), qq = quad_bcast<1>(t), zz3 = quad_bcast<2>(t);
    const FqU x3 = usub<F>(uadd<F>(rr, pppn), uadd<F>(qq, qq));
    // stage 4: R' (X3 - Q) | S1 PPP' | ZZZ12 PPP'
    t = umul<F>(quad_sel(q, rn, s1, zzz12, zzz12), quad_sel(q, usub<F>(x3, qq), pppn, pppn, pppn));
    const FqU y3a = quad_bcast<0>(t), y3b = quad_bcast<1>(t), z3 = quad_bcast<2>(t);
    G1XyzzU o;
    o.x = x3;
    o.y = uadd<F>(y3a, y3b);
    o.zz = zz3;
    o.zzz = usub<F>(u_zero<F>(), z3);
    o.inf = false;
    // behind the last cross-lane move: two finite operands with the same x coordinate (rare) are redone exactly by every lane alone
    if (!a.inf && !b.inf && u_is_zero_mod_p<F>(zz3)) {
        if (fqu_is_zero(pn)) o = fqu_is_zero(rn) ? g1u_dbl(a) : g1u_inf();
    }
    return g1u_select(a.inf, b, g1u_select(b.inf, a, o));
}

// internal accumulator -> stored XYZZ (canonical 32-bit Montgomery limbs)
__device__ __forceinline__ G1Xyzz g1u_to_std(const G1XyzzU &a) {
    using F = Fq381;
    if (a.inf) return g1_xyzz_inf();
    G1Xyzz r;
    r.x = u_to_std<F>(a.x);
    r.y = u_to_std<F>(a.y);
    r.zz = u_to_std<F>(a.zz);
    r.zzz = u_to_std<F>(a.zzz);
    return r;
}

}  // namespace zk
