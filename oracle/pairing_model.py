"""BLS12-381 optimal ate pairing as plain Python big-int arithmetic -- TEST INFRASTRUCTURE ONLY (oracle/).

Independent restatement used to check the product's host-side KZG verifier (csrc/pairing.h): this model works in the
obvious-but-slow way -- G2 points are untwisted into E(Fq12), the Miller loop runs with generic Fq12 arithmetic and
affine slopes, and the final exponentiation is one modular power by (p^12 - 1) / r -- whereas the product keeps G2 on the
twist over Fq2, evaluates sparse lines, shares one Miller accumulator across all pairings and splits the final
exponentiation.  Both must give the same element of GT.

Restated from the published definition of the curve (ark-bls12-381 0.5.0 [ext], absent from /root/reference):
  p, r as in SURVEY.md Appendix A; x = -0xd201000000010000; E: y^2 = x^3 + 4 over Fq; E': y^2 = x^3 + 4 (1 + u) over
  Fq2 = Fq[u]/(u^2 + 1); Fq6 = Fq2[v]/(v^3 - (1 + u)); Fq12 = Fq6[w]/(w^2 - v).
Reference call sites: multilinear_kzg/src/multilinear_kzg.rs:131-158 (verify), trusted_setup.rs:62-72 (g2 powers).
Parity unpinned by the reference (it only round-trips pairings); pinned here by: generator on curve and of order r,
bilinearity e(aP, bQ) = e(P, Q)^(ab), non-degeneracy, and the KZG identity of multilinear_kzg.rs:216-303.
"""
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
X_ABS = 0xd201000000010000          # the curve parameter is -X_ABS

G1 = (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
      0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)
G2 = ((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
       0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
      (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
       0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be))


# ---- Fq2 = Fq[u] / (u^2 + 1): pairs (c0, c1) -----------------------------------------------------------------------
def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a): return ((-a[0]) % P, (-a[1]) % P)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2_scalar(a, k): return (a[0] * k % P, a[1] * k % P)


def f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * d % P, (-a[1]) * d % P)


F2_ZERO, F2_ONE, XI = (0, 0), (1, 0), (1, 1)


# ---- polynomials over Fq2: Fq12 as Fq2[w] / (w^6 - xi), coefficient lists of length 6 (w^2 = v, v^3 = xi) ----------------
def f12_add(a, b): return [f2_add(x, y) for x, y in zip(a, b)]
def f12_sub(a, b): return [f2_sub(x, y) for x, y in zip(a, b)]


def f12_mul(a, b):
    t = [F2_ZERO] * 11
    for i, x in enumerate(a):
        if x == F2_ZERO:
            continue
        for j, y in enumerate(b):
            t[i + j] = f2_add(t[i + j], f2_mul(x, y))
    for k in range(10, 5, -1):                        # w^k = xi w^(k-6)
        t[k - 6] = f2_add(t[k - 6], f2_mul(t[k], XI))
    return t[:6]


F12_ONE = [F2_ONE] + [F2_ZERO] * 5


def f12_from_fq(x): return [(x % P, 0)] + [F2_ZERO] * 5


def f12_pow(a, e):
    out, base = F12_ONE, a
    while e:
        if e & 1:
            out = f12_mul(out, base)
        base = f12_mul(base, base)
        e >>= 1
    return out


def f12_inv(a):
    """a^-1 by linear algebra-free route: a^(p^12 - 2) would be too slow; use the norm chain Fq12 -> Fq6 -> Fq2 -> Fq.
    Written with the conjugation over Fq6 (w -> -w) and a direct 3x3 solve over Fq2 for the cubic extension."""
    # a = A(v) + B(v) w with A = (a0, a2, a4), B = (a1, a3, a5) in Fq6 = Fq2[v]/(v^3 - xi)
    A, B = [a[0], a[2], a[4]], [a[1], a[3], a[5]]
    n = f6_sub(f6_mul(A, A), f6_mul_by_v(f6_mul(B, B)))          # A^2 - v B^2
    ni = f6_inv(n)
    A2, B2 = f6_mul(A, ni), f6_neg(f6_mul(B, ni))
    return [A2[0], B2[0], A2[1], B2[1], A2[2], B2[2]]


def f6_add(a, b): return [f2_add(x, y) for x, y in zip(a, b)]
def f6_sub(a, b): return [f2_sub(x, y) for x, y in zip(a, b)]
def f6_neg(a): return [f2_neg(x) for x in a]


def f6_mul(a, b):
    t = [F2_ZERO] * 5
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            t[i + j] = f2_add(t[i + j], f2_mul(x, y))
    return [f2_add(t[0], f2_mul(t[3], XI)), f2_add(t[1], f2_mul(t[4], XI)), t[2]]


def f6_mul_by_v(a): return [f2_mul(a[2], XI), a[0], a[1]]


def f6_inv(a):
    a0, a1, a2 = a
    A = f2_sub(f2_mul(a0, a0), f2_mul(XI, f2_mul(a1, a2)))
    B = f2_sub(f2_mul(XI, f2_mul(a2, a2)), f2_mul(a0, a1))
    C = f2_sub(f2_mul(a1, a1), f2_mul(a0, a2))
    F = f2_add(f2_mul(a0, A), f2_mul(XI, f2_add(f2_mul(a2, B), f2_mul(a1, C))))
    Fi = f2_inv(F)
    return [f2_mul(A, Fi), f2_mul(B, Fi), f2_mul(C, Fi)]


# ---- curves (affine, None = infinity) --------------------------------------------------------------------------------------
def g1_add(a, b):
    if a is None: return b
    if b is None: return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0: return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x = (lam * lam - a[0] - b[0]) % P
    return (x, (lam * (a[0] - x) - a[1]) % P)


def g1_mul(a, k):
    out = None
    while k:
        if k & 1: out = g1_add(out, a)
        a = g1_add(a, a)
        k >>= 1
    return out


def g1_neg(a): return None if a is None else (a[0], (-a[1]) % P)


def g2_add(a, b):
    if a is None: return b
    if b is None: return a
    if a[0] == b[0]:
        if f2_add(a[1], b[1]) == F2_ZERO: return None
        lam = f2_mul(f2_scalar(f2_mul(a[0], a[0]), 3), f2_inv(f2_scalar(a[1], 2)))
    else:
        lam = f2_mul(f2_sub(b[1], a[1]), f2_inv(f2_sub(b[0], a[0])))
    x = f2_sub(f2_sub(f2_mul(lam, lam), a[0]), b[0])
    return (x, f2_sub(f2_mul(lam, f2_sub(a[0], x)), a[1]))


def g2_mul(a, k):
    out = None
    while k:
        if k & 1: out = g2_add(out, a)
        a = g2_add(a, a)
        k >>= 1
    return out


def g2_neg(a): return None if a is None else (a[0], f2_neg(a[1]))
def g2_on_curve(a): return f2_mul(a[1], a[1]) == f2_add(f2_mul(a[0], f2_mul(a[0], a[0])), f2_scalar(XI, 4))
def g1_on_curve(a): return (a[1] * a[1] - a[0] ** 3 - 4) % P == 0


# ---- pairing -------------------------------------------------------------------------------------------------------------------
def untwist(q):
    """E'(Fq2) -> E(Fq12): (x', y') -> (x' / w^2, y' / w^3).  w^-2 = w^4 / xi, w^-3 = w^3 / xi."""
    xi_inv = f2_inv(XI)
    x = [F2_ZERO] * 6
    y = [F2_ZERO] * 6
    x[4] = f2_mul(q[0], xi_inv)
    y[3] = f2_mul(q[1], xi_inv)
    return x, y


def miller_loop(p1, q2):
    """f_{|x|, Q}(P) with Q untwisted; vertical lines omitted (they lie in a proper subfield)."""
    if p1 is None or q2 is None:
        return F12_ONE
    xq, yq = untwist(q2)
    xp, yp = f12_from_fq(p1[0]), f12_from_fq(p1[1])
    xt, yt = xq, yq
    f = F12_ONE
    three, two = f12_from_fq(3), f12_from_fq(2)
    for bit in bin(X_ABS)[3:]:
        lam = f12_mul(f12_mul(three, f12_mul(xt, xt)), f12_inv(f12_mul(two, yt)))
        line = f12_sub(f12_sub(yp, yt), f12_mul(lam, f12_sub(xp, xt)))
        f = f12_mul(f12_mul(f, f), line)
        x3 = f12_sub(f12_mul(lam, lam), f12_add(xt, xt))
        yt = f12_sub(f12_mul(lam, f12_sub(xt, x3)), yt)
        xt = x3
        if bit == "1":
            lam = f12_mul(f12_sub(yq, yt), f12_inv(f12_sub(xq, xt)))
            line = f12_sub(f12_sub(yp, yt), f12_mul(lam, f12_sub(xp, xt)))
            f = f12_mul(f, line)
            x3 = f12_sub(f12_sub(f12_mul(lam, lam), xt), xq)
            yt = f12_sub(f12_mul(lam, f12_sub(xt, x3)), yt)
            xt = x3
    # the parameter is negative: f_{-|x|} = 1 / f_{|x|} up to vertical lines; conjugation over Fq6 is that inverse after
    # the final exponentiation (w -> -w)
    return [f[0], f2_neg(f[1]), f[2], f2_neg(f[3]), f[4], f2_neg(f[5])]


FINAL_EXP = (P ** 12 - 1) // R


def final_exponentiation(f): return f12_pow(f, FINAL_EXP)
def pairing(p1, q2): return final_exponentiation(miller_loop(p1, q2))


def pairing_product_is_one(pairs):
    f = F12_ONE
    for p1, q2 in pairs:
        f = f12_mul(f, miller_loop(p1, q2))
    return final_exponentiation(f) == F12_ONE


# ---- multilinear KZG verifier: multilinear_kzg.rs:131-158 ----------------------------------------------------------------------------
def kzg_setup_g2(taus):                                  # trusted_setup.rs:62-72
    return [g2_mul(G2, t % R) for t in taus]


def kzg_verify(commitment, opening_values, evaluation, proofs, g2_powers):
    """e(C - [v] G1, G2) == prod_i e(pi_i, [tau_i] G2 - [x_i] G2)"""
    assert len(opening_values) == len(proofs)            # :137-141
    lhs = g1_add(commitment, g1_neg(g1_mul(G1, evaluation % R)))
    pairs = [(lhs, G2)]
    for i, tau_g2 in enumerate(g2_powers):               # :149
        q = g2_add(tau_g2, g2_neg(g2_mul(G2, opening_values[i] % R)))
        pairs.append((g1_neg(proofs[i]), q))
    return pairing_product_is_one(pairs)
