"""GPU parity: G1 MSM (Pippenger), trusted-setup G1 powers, multilinear-KZG commit and open vs the
oracle's NAIVE restatement of multilinear_kzg.rs (group elements compared as affine x, y), plus the
algebraic identities that replace the reference's pairing round-trips."""
import ctypes as C
import random

import numpy as np
import pytest

import __graft_entry__ as G
from oracle import oracle as O
from oracle import pymodel as M

pytestmark = pytest.mark.gpu
R = M.R


@pytest.fixture(scope="module")
def zk():
    zk = G.import_package()
    from zkmle_amd import _lib
    _lib.check(zk.lib().zk_init(0))
    return zk


def rand_fr(zk, n, seed):
    t = np.zeros((n, 4), np.uint64)
    assert zk.lib().zk_host_fill_random(0, seed, 0, n, t.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    return t


def test_generator_matches_oracle(zk):
    g = zk.kzg.g1_generator()
    assert np.array_equal(g, O.g1_generator()) and zk.kzg.g1_is_on_curve(g)


def test_lagrange_basis_kats(zk, ref_kats):
    for k in ref_kats["kzg_lagrange_basis"]:                  # trusted_setup.rs:93-118
        got = zk.kzg.compute_lagrange_basis(zk.from_ints(0, k["taus"])).to_ints()
        assert got == [v % R for v in k["expect"]]
    taus = rand_fr(zk, 10, 77)
    assert np.array_equal(zk.kzg.compute_lagrange_basis(taus).evaluated_values, O.kzg_lagrange_basis(taus))


def test_setup_commit_open_reference_cases(zk, ref_kats, derived_kats):
    derived = {d["src"]: d for d in derived_kats["kzg"]}
    for k in ref_kats["kzg_roundtrip"]:                       # multilinear_kzg.rs:216-303
        taus = zk.from_ints(0, k["taus"])
        setup = zk.TrustedSetup.initialize_setup(taus)
        want_pts = O.kzg_setup_g1(taus)
        assert np.array_equal(setup.g1_powers_of_tau.points(), want_pts), k["src"]
        vals = zk.from_ints(0, k["values"])
        poly = zk.MultilinearPolynomial(0, vals)
        commitment = zk.MultilinearKZG.commit_to_polynomial(poly, setup)
        assert np.array_equal(commitment, O.kzg_commit(vals, want_pts))
        opening = zk.from_ints(0, k["opening"])
        proof = zk.MultilinearKZG.open_and_prove(poly, setup, opening)
        ev, proofs = O.kzg_open(vals, want_pts, opening)
        assert np.array_equal(proof.evaluation, ev) and np.array_equal(proof.proofs, proofs)
        d = derived[k["src"]]
        assert zk.to_ints(0, proof.evaluation) == [d["f_open"] % R]
        if "commit_x" in d:
            assert O.g1_affine_ints(commitment) == (int(d["commit_x"], 16), int(d["commit_y"], 16))
        # commitment == [f(tau)] G  (what the pairing check states when tau is known)
        assert O.g1_affine_ints(commitment) == M.g1_mul(M.G1, d["f_tau"] % R)
        # ... and the pairing check itself: MultilinearKZG::verify, the reference's own round trip (:216-303)
        assert zk.MultilinearKZG.verify(setup, commitment, opening, proof) is True
        wrong = zk.MultilinearKZGProof(zk.from_ints(0, [(d["f_open"] + 1) % R])[0], proof.proofs)
        assert zk.MultilinearKZG.verify(setup, commitment, opening, wrong) is False


def test_kzg_length_asserts(zk):
    setup = zk.TrustedSetup.initialize_setup(zk.from_ints(0, [5, 2]))
    with pytest.raises(zk.ReferencePanic):                    # multilinear_kzg.rs:29-33
        zk.MultilinearKZG.commit_to_polynomial(zk.MultilinearPolynomial.from_ints(0, [1, 2]), setup)
    with pytest.raises(zk.ReferencePanic):                    # :55-59
        zk.MultilinearKZG.open_and_prove(zk.MultilinearPolynomial.from_ints(0, [1, 2, 3, 4]), setup, zk.from_ints(0, [7]))


def test_zero_scalars_and_infinity_bases(zk):
    taus = zk.from_ints(0, [1, 0, 7])                         # Lagrange coefficients 0 -> bases at infinity
    setup = zk.TrustedSetup.initialize_setup(taus)
    pts = setup.g1_powers_of_tau.points()
    assert np.array_equal(pts, O.kzg_setup_g1(taus))
    assert sum(1 for p in pts if not p.any()) == 6
    vals = zk.from_ints(0, [3, 0, 9, 1, 0, 0, 5, 11])
    c = zk.MultilinearKZG.commit_to_polynomial(zk.MultilinearPolynomial(0, vals), setup)
    assert np.array_equal(c, O.kzg_commit(vals, pts))
    zero = zk.MultilinearKZG.commit_to_polynomial(zk.MultilinearPolynomial.from_ints(0, [0] * 8), setup)
    assert not zero.any()                                     # the point at infinity


@pytest.mark.parametrize("n,c", [(1, 0), (2, 0), (3, 4), (64, 0), (64, 2), (200, 5), (1 << 10, 0), (1 << 10, 16), (1 << 10, 9)])
def test_msm_vs_naive_oracle(zk, n, c):
    """random scalars, random (distinct) bases; every supported window size agrees with double-and-add"""
    scalars = rand_fr(zk, n, 1000 + n)
    a, d = rand_fr(zk, 2, 5)
    bases = zk.G1Bases.synthetic(n, a, d)
    pts = bases.points()
    # the synthetic generator itself: P_i = [a + i d] G
    g = O.g1_generator()
    ai, di = O.to_ints(O.FR381, np.stack([a, d]))
    for i in sorted({0, min(1, n - 1), n - 1}):
        assert O.g1_affine_ints(pts[i]) == M.g1_mul(M.G1, (ai + i * di) % R)
    st = zk.MultilinearPolynomial.vector(0, scalars)
    got, stats = zk.kzg.msm(st, bases, window_bits=c, with_stats=True)
    want = O.kzg_commit(scalars, pts)
    assert np.array_equal(got, want), stats
    assert stats["terms"] == n


def test_msm_skewed_and_degenerate_inputs(zk):
    """all-equal scalars (one bucket per window gets everything -> segments), tiny scalars, scalar r-1,
    and repeated bases (forces the P = Q doubling path and P = -Q cancellation inside buckets)"""
    n = 1 << 9
    a, d = rand_fr(zk, 2, 9)
    bases = zk.G1Bases.synthetic(n, a, d)
    pts = bases.points()
    MP = zk.MultilinearPolynomial
    for vals in ([12345678901234567890123] * n, [1] * n, [R - 1] * n, [0] * (n - 1) + [7], [i % 3 for i in range(n)]):
        sc = zk.from_ints(0, vals)
        assert np.array_equal(zk.kzg.msm(MP(0, sc), bases), O.kzg_commit(sc, pts))
    rep = np.tile(pts[:2], (n // 2, 1))                        # only two distinct points: P, Q, P, Q, ...
    rep[5] = O.g1_neg(rep[4 + 1])                              # and one negated copy
    repb = zk.G1Bases(rep)
    sc = zk.from_ints(0, [5] * n)
    assert np.array_equal(zk.kzg.msm(MP(0, sc), repb, window_bits=4), O.kzg_commit(sc, rep))
    sc = rand_fr(zk, n, 31)
    assert np.array_equal(zk.kzg.msm(MP(0, sc), repb), O.kzg_commit(sc, rep))


def test_msm_linear_checksum_2p16(zk):
    """O(N) field check of a large MSM: bases [a + i d] G  =>  MSM == [a sum s_i + d sum i s_i] G"""
    n = 1 << 16
    a, d = rand_fr(zk, 2, 12)
    bases = zk.G1Bases.synthetic(n, a, d)
    scalars = zk.MultilinearPolynomial.random(0, n, 0x5EED0003)
    got = zk.kzg.msm(scalars, bases)
    s = scalars.to_ints()
    ai, di = O.to_ints(O.FR381, np.stack([a, d]))
    k = (ai * sum(s) + di * sum(i * v for i, v in enumerate(s))) % R
    assert O.g1_affine_ints(got) == M.g1_mul(M.G1, k)


def test_msm_config3_size_2p20(zk):
    """BASELINE config 3: 2^20-term MSM; checked with the O(N) linear identity for bases [a + i d] G"""
    n = 1 << 20
    a, d = rand_fr(zk, 2, 21)
    bases = zk.G1Bases.synthetic(n, a, d)
    scalars = zk.MultilinearPolynomial.random(0, n, 0x5EED0003)
    got, stats = zk.kzg.msm(scalars, bases, with_stats=True)
    assert stats["window_bits"] == 16 and stats["terms"] == n
    s = scalars.to_ints()
    ai, di = O.to_ints(O.FR381, np.stack([a, d]))
    k = (ai * sum(s) + di * sum(i * v for i, v in enumerate(s))) % R
    assert O.g1_affine_ints(got) == M.g1_mul(M.G1, k)
    # every window size lands on the same group element
    for c in (11, 14):
        assert np.array_equal(zk.kzg.msm(scalars, bases, window_bits=c), got)


def test_commit_open_n16(zk):
    """KZG commit + open of a 2^16 table: evaluation, proof count and the identity sum_i Q_i(tau)(tau_i - x_i) = f(tau) - v
    with the prover's own group elements (pi_i = [Q_i(tau)] G checked through the scalar identity on G)"""
    rng = random.Random(16)
    nv = 16
    taus_i = [rng.randrange(R) for _ in range(nv)]
    taus = zk.from_ints(0, taus_i)
    setup = zk.TrustedSetup.initialize_setup(taus)
    poly = zk.MultilinearPolynomial.random(0, 1 << nv, 1616)
    vals = poly.evaluated_values
    f_tau = O.to_ints(O.FR381, O.evaluate(O.FR381, vals, taus))[0]
    c = zk.MultilinearKZG.commit_to_polynomial(poly, setup)
    assert O.g1_affine_ints(c) == M.g1_mul(M.G1, f_tau)
    opening_i = [rng.randrange(R) for _ in range(nv)]
    opening = zk.from_ints(0, opening_i)
    proof = zk.MultilinearKZG.open_and_prove(poly, setup, opening)
    v = O.to_ints(O.FR381, O.evaluate(O.FR381, vals, opening))[0]
    assert zk.to_ints(0, proof.evaluation) == [v] and len(proof.proofs) == nv
    # sum_i (tau_i - x_i) * pi_i  ==  [f(tau) - v] G   (the pairing equation projected to G1, tau known)
    acc = np.zeros(12, np.uint64)
    for i in range(nv):
        acc = O.g1_add(acc, O.g1_mul_fr(proof.proofs[i], O.from_ints(O.FR381, [(taus_i[i] - opening_i[i]) % R])[0]))
    assert O.g1_affine_ints(acc) == M.g1_mul(M.G1, (f_tau - v) % R)
    # 17 pairings against the G2 powers (host): accepts the proof, rejects a swapped pair of quotient commitments
    assert zk.MultilinearKZG.verify(setup, c, opening, proof) is True
    swapped = proof.proofs.copy()
    swapped[[0, 1]] = swapped[[1, 0]]
    assert zk.MultilinearKZG.verify(setup, c, opening, zk.MultilinearKZGProof(proof.evaluation, swapped)) is False


def test_commit_open_random_n10(zk):
    """larger random KZG instance: commit == [f(tau)] G, proofs == [Q_i(tau)] G, identity sum"""
    rng = random.Random(4)
    nv = 10
    taus_i = [rng.randrange(R) for _ in range(nv)]
    taus = zk.from_ints(0, taus_i)
    setup = zk.TrustedSetup.initialize_setup(taus)
    poly = zk.MultilinearPolynomial.random(0, 1 << nv, 99)
    vals = poly.evaluated_values
    f_tau = O.to_ints(O.FR381, O.evaluate(O.FR381, vals, taus))[0]
    c = zk.MultilinearKZG.commit_to_polynomial(poly, setup)
    assert O.g1_affine_ints(c) == M.g1_mul(M.G1, f_tau)
    opening_i = [rng.randrange(R) for _ in range(nv)]
    opening = zk.from_ints(0, opening_i)
    proof = zk.MultilinearKZG.open_and_prove(poly, setup, opening)
    v = O.to_ints(O.FR381, O.evaluate(O.FR381, vals, opening))[0]
    assert zk.to_ints(0, proof.evaluation) == [v]
    qs = O.kzg_quotients(vals, opening)
    acc = 0
    for i, q in enumerate(qs):
        q_tau = O.to_ints(O.FR381, O.evaluate(O.FR381, q, taus[i + 1:]))[0] if len(q) > 1 else O.to_ints(O.FR381, q)[0]
        assert O.g1_affine_ints(proof.proofs[i]) == M.g1_mul(M.G1, q_tau), i
        acc = (acc + q_tau * (taus_i[i] - opening_i[i])) % R
    assert acc == (f_tau - v) % R


def test_msm_config5_size_2p24(zk):
    """BASELINE config 5: the 2^24-term MSM, checked with the O(N) linear identity for bases [a + i d] G; the two field sums
    over 16.7M scalars are taken with vectorised 16-bit-piece arithmetic on the canonical limbs (exact)."""
    from zkmle_amd import _lib as L
    n = 1 << 24
    a, d = rand_fr(zk, 2, 24)
    bases = zk.G1Bases.synthetic(n, a, d)
    scalars = zk.MultilinearPolynomial.random(0, n, 0x5EED0003)
    got, stats = zk.kzg.msm(scalars, bases, with_stats=True)
    assert stats["window_bits"] == 16 and stats["terms"] == n and stats["windows"] == 16
    mont = scalars.evaluated_values
    canon = np.zeros_like(mont)
    L.check(L.lib().zk_vec_to_canonical(0, L.p64(mont), n, L.p64(canon)))
    idx = np.arange(n, dtype=np.uint64)
    s_sum, is_sum = 0, 0
    step = 1 << 20
    for k in range(4):
        for j in range(4):
            piece = (canon[:, k] >> np.uint64(16 * j)) & np.uint64(0xFFFF)
            shift = 64 * k + 16 * j
            s_sum += int(piece.sum(dtype=np.uint64)) << shift                   # < 2^40
            acc = 0
            for lo in range(0, n, step):                                        # 2^20 terms below 2^40 each: < 2^60
                acc += int((idx[lo:lo + step] * piece[lo:lo + step]).sum(dtype=np.uint64))
            is_sum += acc << shift
    ai, di = O.to_ints(O.FR381, np.stack([a, d]))
    k = (ai * s_sum + di * is_sum) % R
    assert O.g1_affine_ints(got) == M.g1_mul(M.G1, k)


def test_open_large_levels_on_side_streams_2p21(zk):
    """2^21 terms: one level MSM (2^20 terms) plus the batched pass of the small levels run on two host threads with their own
    streams (zk_kzg_open); the pairing check and the projected identity accept the proof, and it equals the one-thread proof"""
    rng = random.Random(21)
    nv = 21
    taus_i = [rng.randrange(R) for _ in range(nv)]
    taus = zk.from_ints(0, taus_i)
    setup = zk.TrustedSetup.initialize_setup(taus)
    poly = zk.MultilinearPolynomial.random(0, 1 << nv, 2121)
    c = zk.MultilinearKZG.commit_to_polynomial(poly, setup)
    opening_i = [rng.randrange(R) for _ in range(nv)]
    opening = zk.from_ints(0, opening_i)
    proof = zk.MultilinearKZG.open_and_prove(poly, setup, opening)
    again = zk.MultilinearKZG.open_and_prove(poly, setup, opening)
    assert np.array_equal(proof.proofs, again.proofs) and np.array_equal(proof.evaluation, again.evaluation)
    assert zk.MultilinearKZG.verify(setup, c, opening, proof) is True
    bad = proof.proofs.copy()
    bad[0] = proof.proofs[1]
    assert zk.MultilinearKZG.verify(setup, c, opening, zk.MultilinearKZGProof(proof.evaluation, bad)) is False
    # window-shifted copies of the setup and of the opening key's large levels (zk_g1_bases_precompute, zk_kzg_opening_key_precompute):
    # the commitment and every proof point are the same group elements
    setup.precompute_for_commits()
    setup.precompute_for_opens(min_points=1 << 16)
    assert np.array_equal(zk.MultilinearKZG.commit_to_polynomial(poly, setup), c)
    pre = zk.MultilinearKZG.open_and_prove(poly, setup, opening)
    assert np.array_equal(pre.proofs, proof.proofs) and np.array_equal(pre.evaluation, proof.evaluation)


def test_msm_and_open_on_a_nonblocking_user_stream(zk):
    """every launch of the MSM pipeline follows the thread's current stream (zk_set_stream): a non-blocking stream does not
    synchronise with the null stream, so a stray null-stream launch would read unsorted indices"""
    import torch
    n = 1 << 13
    scalars = rand_fr(zk, n, 77)
    a, d = rand_fr(zk, 2, 78)
    bases = zk.G1Bases.synthetic(n, a, d)
    want = O.kzg_commit(scalars[:256], bases.points()[:256])
    st = torch.cuda.Stream()
    L = zk.lib()
    L.zk_set_stream.argtypes = [C.c_void_p]
    assert L.zk_set_stream(C.c_void_p(st.cuda_stream)) == 0
    try:
        small = zk.G1Bases(bases.points()[:256])
        got = zk.kzg.msm(zk.MultilinearPolynomial.vector(0, scalars[:256]), small)
        assert np.array_equal(got, want)
        ai, di = O.to_ints(O.FR381, np.stack([a, d]))
        s_int = O.to_ints(O.FR381, scalars)
        k = (ai * sum(s_int) + di * sum(i * s for i, s in enumerate(s_int))) % R
        got = zk.kzg.msm(zk.MultilinearPolynomial.vector(0, scalars), bases)
        assert O.g1_affine_ints(got) == M.g1_mul(M.G1, k)
    finally:
        assert L.zk_set_stream(None) == 0


def test_setup_with_infinite_points_2p21(zk):
    """tau_0 = 1 and tau_3 = 0 make three quarters of the Lagrange coefficients zero: the long-batch normalisation, the pre-summed
    levels and the MSMs all see points at infinity at a size that takes the large-array kernels"""
    rng = random.Random(2121)
    nv = 21
    taus_i = [rng.randrange(R) for _ in range(nv)]
    taus_i[0], taus_i[3] = 1, 0
    taus = zk.from_ints(0, taus_i)
    setup = zk.TrustedSetup.initialize_setup(taus)
    pts = setup.g1_powers_of_tau.points()
    inf = ~pts.any(axis=1)
    assert int(inf.sum()) == 3 * (1 << (nv - 2))
    assert not inf[(1 << (nv - 1)) + 5] and inf[5]                     # index bit (n-1) = variable 0 must be 1, variable 3 must be 0
    poly = zk.MultilinearPolynomial.random(0, 1 << nv, 777)
    f_tau = O.to_ints(O.FR381, O.evaluate(O.FR381, poly.evaluated_values, taus))[0]
    c = zk.MultilinearKZG.commit_to_polynomial(poly, setup)
    assert O.g1_affine_ints(c) == M.g1_mul(M.G1, f_tau)
    opening = zk.from_ints(0, [rng.randrange(R) for _ in range(nv)])
    proof = zk.MultilinearKZG.open_and_prove(poly, setup, opening)
    assert zk.MultilinearKZG.verify(setup, c, opening, proof) is True
