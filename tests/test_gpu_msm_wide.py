"""GPU parity of the Pippenger MSM beyond 16-bit windows and on random bases (multilinear_kzg.rs:37-42 is a naive sum; every
variant here must return the same group element):
  * windows of 17 .. 24 bits (csrc/msm_sort_wide.cuh: most-significant-digit-first counting sort in three levels),
  * precomputed window-shifted bases (zk_g1_bases_precompute: every window feeds ONE bucket set),
  * 2^18 / 2^20 terms on RANDOM bases -- a real trusted setup and a shuffled copy with duplicates, negations and points at
    infinity -- against the oracle's own CPU Pippenger (oracle/g1.c orc_msm_pippenger: unsigned windows, OpenMP; an independent
    bucket method, itself checked against the naive sum in tests/test_oracle_kzg.py)."""
import ctypes as C

import numpy as np
import pytest

import __graft_entry__ as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zk():
    zk = G.import_package()
    from zkmle_amd import _lib
    _lib.check(zk.lib().zk_init(0))
    return zk


@pytest.fixture(autouse=True)
def release_cached_scratch(zk):
    """wide windows reduce over up to 2^23 buckets per set: tens of GB of per-call scratch that the caching pool would keep"""
    yield
    zk.lib().zk_release_cached_memory()


def rand_fr(zk, n, seed):
    t = np.zeros((n, 4), np.uint64)
    assert zk.lib().zk_host_fill_random(0, seed, 0, n, t.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    return t


def expected(scalars, pts):
    n = len(scalars)
    return O.kzg_commit(scalars, pts) if n <= 1024 else O.msm_pippenger(scalars, pts, window_bits=10, slices=8)


@pytest.mark.parametrize("n,c", [(1, 17), (3, 20), (300, 17), (1 << 10, 18), (1 << 10, 20), (1 << 10, 22), (777, 24), (1 << 13, 19), (1 << 13, 21),
                                 (1 << 15, 23)])
def test_wide_windows_vs_oracle(zk, n, c):
    scalars = rand_fr(zk, n, 4000 + n + c)
    a, d = rand_fr(zk, 2, 6)
    bases = zk.G1Bases.synthetic(n, a, d)
    got, stats = zk.kzg.msm(zk.MultilinearPolynomial.vector(0, scalars), bases, window_bits=c, with_stats=True)
    assert stats["window_bits"] == c and stats["windows"] == (256 + c - 1) // c and stats["terms"] == n
    assert np.array_equal(got, expected(scalars, bases.points())), stats


def test_wide_windows_skewed_and_degenerate(zk):
    """all-equal scalars: every entry of a window lands in ONE bucket, i.e. one level-3 group far larger than its LDS tile (the
    count-first path) and one bucket cut into many segments; plus tiny scalars, r - 1, zeros, repeated and negated bases"""
    R = O.modulus(O.FR381)
    n = 1 << 14
    a, d = rand_fr(zk, 2, 19)
    bases = zk.G1Bases.synthetic(n, a, d)
    pts = bases.points()
    MP = zk.MultilinearPolynomial
    for c in (17, 20):
        for vals in ([12345678901234567890123] * n, [1] * n, [R - 1] * n, [0] * (n - 1) + [7], [i % 3 for i in range(n)]):
            sc = zk.from_ints(0, vals)
            assert np.array_equal(zk.kzg.msm(MP.vector(0, sc), bases, window_bits=c), expected(sc, pts)), (c, vals[:2])
    rep = np.tile(pts[:2], (n // 2, 1))                        # only two distinct points: P, Q, P, Q, ...
    rep[5] = O.g1_neg(rep[5])
    rep[8] = 0                                                 # and a point at infinity
    repb = zk.G1Bases(rep)
    sc = rand_fr(zk, n, 32)
    assert np.array_equal(zk.kzg.msm(MP.vector(0, sc), repb, window_bits=18), expected(sc, rep))


@pytest.mark.parametrize("n,c", [(5, 0), (1000, 13), (1000, 16), (1 << 12, 20), (1 << 13, 22), (1 << 14, 0)])
def test_precomputed_window_copies_give_the_same_point(zk, n, c):
    scalars = rand_fr(zk, n, 5000 + n + c)
    a, d = rand_fr(zk, 2, 7)
    pts = zk.G1Bases.synthetic(n, a, d).points()
    pts[n // 2] = 0                                            # 2^(c w) * infinity stays infinity
    pts[n // 3] = pts[0]
    want = expected(scalars, pts)
    bases = zk.G1Bases(pts)
    sc = zk.MultilinearPolynomial.vector(0, scalars)
    plain = zk.kzg.msm(sc, bases)
    assert np.array_equal(plain, want)
    used = bases.precompute(c)
    assert used == (c or used) and used >= 9
    got, stats = zk.kzg.msm(sc, bases, with_stats=True)        # window_bits = 0: the precomputed copies are used
    assert stats["window_bits"] == used and np.array_equal(got, want), stats
    other = 11 if used != 11 else 12
    assert np.array_equal(zk.kzg.msm(sc, bases, window_bits=other), want)      # another window size: the plain path
    assert bases.precompute(used) == used                      # idempotent
    assert np.array_equal(bases.points(), pts)                 # the stored points are untouched


@pytest.mark.parametrize("logn", [18, 20])
def test_large_msm_on_random_bases_vs_cpu_pippenger(zk, logn):
    """the sizes the structured-bases identity covers, on bases with no structure at all"""
    n = 1 << logn
    taus = rand_fr(zk, logn, 6000 + logn)
    setup = zk.TrustedSetup.initialize_setup(taus)             # [L_i(tau)] G: a real setup (trusted_setup.rs:51-60)
    pts = setup.g1_powers_of_tau.points()
    scalars = rand_fr(zk, n, 6100 + logn)
    want = O.msm_pippenger(scalars, pts, window_bits=12, slices=8)
    sc = zk.MultilinearPolynomial.vector(0, scalars)
    got, stats = zk.kzg.msm(sc, setup.g1_powers_of_tau, with_stats=True)
    assert np.array_equal(got, want), stats
    for c in (16, 20):                                         # the narrow and the wide sort
        assert np.array_equal(zk.kzg.msm(sc, setup.g1_powers_of_tau, window_bits=c), want), c
    # a shuffled copy with duplicates, negations and points at infinity sprinkled in
    rng = np.random.default_rng(logn)
    mixed = pts[rng.permutation(n)]
    idx = rng.choice(n, 3000, replace=False)
    mixed[idx[:1000]] = mixed[idx[1000:2000]]                  # duplicates (P + P inside a bucket when the digits agree)
    for j in idx[2000:2500]:
        mixed[j] = O.g1_neg(mixed[(j + 1) % n])                # P and -P
    mixed[idx[2500:]] = 0                                      # infinity
    small = scalars.copy()
    small[idx[:2000]] = scalars[idx[0]]                        # equal scalars on the duplicated points: same bucket in every window
    want2 = O.msm_pippenger(small, mixed, window_bits=12, slices=8)
    mb = zk.G1Bases(mixed)
    sc2 = zk.MultilinearPolynomial.vector(0, small)
    assert np.array_equal(zk.kzg.msm(sc2, mb), want2)
    assert np.array_equal(zk.kzg.msm(sc2, mb, window_bits=19), want2)
    if logn == 18:
        mb.precompute(0)
        assert np.array_equal(zk.kzg.msm(sc2, mb), want2)
