"""Host-side pairing / KZG verifier (csrc/pairing.h, C ABI zk_pairing*, zk_kzg_setup_g2, zk_kzg_verify) against the independent
big-int model oracle/pairing_model.py.  Host code only: runs without a GPU.  The reference pins nothing here (it only
round-trips pairings, multilinear_kzg.rs:216-303); the pins are the generator checks, bilinearity, and the KZG identity."""
import ctypes as C

import numpy as np
import pytest

import __graft_entry__ as G
from oracle import pairing_model as M
from oracle import pymodel as PM

zkp = G.import_package()
from zkmle_amd import _lib as L   # noqa: E402

RQ = 1 << 384


def lib():
    lb = L.lib()
    if not getattr(lb, "_pairing_declared", False):
        u64p = L.u64p
        lb.zk_g2_generator.argtypes = [u64p]
        lb.zk_g2_is_on_curve.argtypes = [u64p]
        lb.zk_g2_add.argtypes = [u64p, u64p, u64p]
        lb.zk_g2_mul_fr.argtypes = [u64p, u64p, u64p]
        lb.zk_pairing.argtypes = [u64p, u64p, u64p]
        lb.zk_pairing_product_is_one.argtypes = [u64p, u64p, C.c_size_t, C.POINTER(C.c_int)]
        lb.zk_kzg_setup_g2.argtypes = [u64p, C.c_size_t, u64p]
        lb.zk_kzg_verify.argtypes = [u64p, u64p, C.c_size_t, u64p, u64p, C.c_size_t, u64p, C.c_size_t, C.POINTER(C.c_int)]
        lb._pairing_declared = True
    return lb


def fq_limbs(v):
    v = v * RQ % M.P
    return [(v >> (64 * i)) & ((1 << 64) - 1) for i in range(6)]


def fq_int(limbs):
    return sum(int(x) << (64 * i) for i, x in enumerate(limbs)) * pow(RQ, -1, M.P) % M.P


def g1_arr(p):
    return np.array(([0] * 12) if p is None else fq_limbs(p[0]) + fq_limbs(p[1]), np.uint64)


def g2_arr(q):
    return np.array(([0] * 24) if q is None else fq_limbs(q[0][0]) + fq_limbs(q[0][1]) + fq_limbs(q[1][0]) + fq_limbs(q[1][1]), np.uint64)


def g2_from(arr):
    v = [fq_int(arr[6 * i: 6 * i + 6]) for i in range(4)]
    return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))


def fr_arr(k):
    v = k % M.R * (1 << 256) % M.R
    return np.array([(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], np.uint64)


def gt_from(arr):
    return [(fq_int(arr[12 * k: 12 * k + 6]), fq_int(arr[12 * k + 6: 12 * k + 12])) for k in range(6)]


def c_pairing(p, q):
    out = np.zeros(72, np.uint64)
    L.check(lib().zk_pairing(L.p64(g1_arr(p)), L.p64(g2_arr(q)), L.p64(out)))
    return gt_from(out)


def test_g2_generator_and_arithmetic():
    g = np.zeros(24, np.uint64)
    L.check(lib().zk_g2_generator(L.p64(g)))
    assert g2_from(g) == M.G2 and lib().zk_g2_is_on_curve(L.p64(g)) == 1
    for k in (1, 2, 3, 0xdeadbeefcafebabe1234567, M.R - 1, M.R):
        out = np.zeros(24, np.uint64)
        L.check(lib().zk_g2_mul_fr(L.p64(g), L.p64(fr_arr(k)), L.p64(out)))
        assert g2_from(out) == M.g2_mul(M.G2, k % M.R), k
    a, b = M.g2_mul(M.G2, 77), M.g2_mul(M.G2, 1000003)
    for x, y in ((a, b), (a, a), (a, M.g2_neg(a)), (None, b), (a, None)):
        out = np.zeros(24, np.uint64)
        L.check(lib().zk_g2_add(L.p64(g2_arr(x)), L.p64(g2_arr(y)), L.p64(out)))
        assert g2_from(out) == M.g2_add(x, y)


def test_pairing_golden_vectors(derived_kats):
    """the committed GT values (tests/golden/derived_kats.json, made with oracle/pairing_model.py) pin the product's pairing"""
    k = derived_kats["pairing"]
    want = [(int(c0, 16), int(c1, 16)) for c0, c1 in k["e_g1_g2"]]
    assert c_pairing(M.G1, M.G2) == want
    a, b = int(k["a"], 16), int(k["b"], 16)
    want2 = [(int(c0, 16), int(c1, 16)) for c0, c1 in k["e_aG1_bG2"]]
    assert c_pairing(M.g1_mul(M.G1, a), M.g2_mul(M.G2, b)) == want2


def test_pairing_equals_the_model_and_is_bilinear():
    e = c_pairing(M.G1, M.G2)
    assert e == M.pairing(M.G1, M.G2)                      # same element of GT from two different algorithms
    assert e != M.F12_ONE and M.f12_pow(e, M.R) == M.F12_ONE
    a, b = 0x1234567890abcdef1234567, 0xfedcba9876543210fedcba
    pa, qb = M.g1_mul(M.G1, a), M.g2_mul(M.G2, b)
    assert c_pairing(pa, qb) == M.f12_pow(e, a * b % M.R) == M.pairing(pa, qb)
    assert c_pairing(None, M.G2) == M.F12_ONE and c_pairing(M.G1, None) == M.F12_ONE
    ok = C.c_int(0)
    g1s = np.stack([g1_arr(pa), g1_arr(M.g1_neg(M.G1))])
    g2s = np.stack([g2_arr(M.G2), g2_arr(M.g2_mul(M.G2, a))])
    L.check(lib().zk_pairing_product_is_one(L.p64(g1s), L.p64(g2s), 2, C.byref(ok)))
    assert ok.value == 1                                   # e(aG1, G2) e(-G1, aG2) = 1
    g2s[1] = g2_arr(M.g2_mul(M.G2, a + 1))
    L.check(lib().zk_pairing_product_is_one(L.p64(g1s), L.p64(g2s), 2, C.byref(ok)))
    assert ok.value == 0


@pytest.mark.parametrize("nvars", [1, 2, 3])
def test_kzg_verify_roundtrip_and_tampering(nvars):
    """multilinear_kzg.rs:216-303 (commit -> open -> verify is true; a wrong evaluation / proof / point is false), with the
    commitment and proofs computed by the big-int model of commit / open"""
    rng = np.random.default_rng(40 + nvars)
    taus = [int.from_bytes(rng.bytes(40), "little") % M.R for _ in range(nvars)]
    vals = [int.from_bytes(rng.bytes(40), "little") % M.R for _ in range(1 << nvars)]
    point = [int.from_bytes(rng.bytes(40), "little") % M.R for _ in range(nvars)]
    pts = PM.kzg_setup_g1(taus)
    commitment = PM.kzg_commit(vals, pts)
    evaluation, proofs = PM.kzg_open(vals, pts, point)
    g2p = np.zeros((nvars, 24), np.uint64)
    L.check(lib().zk_kzg_setup_g2(L.p64(np.stack([fr_arr(t) for t in taus])), nvars, L.p64(g2p)))
    assert [g2_from(r) for r in g2p] == M.kzg_setup_g2(taus)           # trusted_setup.rs:62-72

    def verify(c, pt, ev, prs):
        ok = C.c_int(-1)
        L.check(lib().zk_kzg_verify(L.p64(g1_arr(c)), L.p64(np.stack([fr_arr(x) for x in pt])), len(pt), L.p64(fr_arr(ev)),
                                    L.p64(np.stack([g1_arr(p) for p in prs])), len(prs), L.p64(g2p), nvars, C.byref(ok)))
        return ok.value

    assert verify(commitment, point, evaluation, proofs) == 1
    assert M.kzg_verify(commitment, point, evaluation, proofs, M.kzg_setup_g2(taus))
    assert verify(commitment, point, (evaluation + 1) % M.R, proofs) == 0
    assert verify(M.g1_add(commitment, M.G1), point, evaluation, proofs) == 0
    bad = list(proofs)
    bad[0] = M.g1_add(bad[0], M.G1) if bad[0] is not None else M.G1
    assert verify(commitment, point, evaluation, bad) == 0
    wrong_point = [(point[0] + 1) % M.R] + point[1:]
    assert verify(commitment, wrong_point, evaluation, proofs) == 0
    with pytest.raises(zkp.ReferencePanic):                              # multilinear_kzg.rs:137-141
        verify(commitment, point + [1], evaluation, proofs)
