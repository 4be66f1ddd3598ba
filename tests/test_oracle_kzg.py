"""Oracle pin 3: BLS12-381 G1 and the multilinear-KZG prover side.

No reference test states any G1 coordinate (parity unpinned): commit / open are pinned
here through (a) two independent restatements (C Jacobian vs Python affine big-int),
(b) the [derived] Appendix-B commitment coordinates, and (c) the algebraic identities that the
reference's pairing round-trips (multilinear_kzg.rs:216-303) are equivalent to when tau is known:
   commit == [f(tau)] G ,  proof_i == [Q_i(tau_{i+1..})] G ,  sum_i Q_i(tau)(tau_i - x_i) == f(tau) - v.
"""
import random

import numpy as np

from oracle import oracle as O
from oracle import pymodel as M

R = M.R


def pt(p):
    return O.g1_affine_ints(p)


def test_generator_and_group_laws():
    g = O.g1_generator()
    assert pt(g) == M.G1 and O.g1_is_on_curve(g)
    assert M.g1_mul(M.G1, R) is None                     # r*G = infinity
    two = O.g1_add(g, g)
    assert pt(two) == M.g1_add(M.G1, M.G1) and O.g1_is_on_curve(two)
    assert pt(O.g1_add(g, O.g1_neg(g))) is None            # P + (-P)
    inf = np.zeros(12, np.uint64)
    assert pt(O.g1_add(inf, g)) == M.G1 and pt(O.g1_add(g, inf)) == M.G1
    rng = random.Random(3)
    for _ in range(6):
        k = rng.randrange(R)
        got = O.g1_mul_fr(g, O.from_ints(O.FR381, [k])[0])
        assert pt(got) == M.g1_mul(M.G1, k)
    assert pt(O.g1_mul_fr(g, O.from_ints(O.FR381, [0])[0])) is None
    assert pt(O.g1_mul_fr(g, O.from_ints(O.FR381, [R - 1])[0])) == pt(O.g1_neg(g))


def _eval_at(values, point):
    return M.evaluate([v % R for v in values], [t % R for t in point], R)


def test_kzg_reference_cases(ref_kats, derived_kats):
    derived = {d["src"]: d for d in derived_kats["kzg"]}
    for k in ref_kats["kzg_roundtrip"]:
        taus, values, opening = k["taus"], k["values"], k["opening"]
        n = len(taus)
        pts = O.kzg_setup_g1(O.from_ints(O.FR381, taus))
        mpts = M.kzg_setup_g1(taus)
        assert [pt(p) for p in pts] == mpts
        vals = O.from_ints(O.FR381, values)
        commit = O.kzg_commit(vals, pts)
        f_tau = _eval_at(values, taus)
        assert pt(commit) == M.kzg_commit(values, mpts) == M.g1_mul(M.G1, f_tau)
        ev, proofs = O.kzg_open(vals, pts, O.from_ints(O.FR381, opening))
        mv, mproofs = M.kzg_open([v % R for v in values], mpts, [x % R for x in opening])
        v = O.to_ints(O.FR381, ev)[0]
        assert v == mv == _eval_at(values, opening)
        assert [pt(p) for p in proofs] == mproofs
        # algebraic identity == what the pairing check e(C - vG, H) = prod e(pi_i, (tau_i - x_i)H) states
        qs = O.kzg_quotients(vals, O.from_ints(O.FR381, opening))
        acc = 0
        for i, q in enumerate(qs):
            q_tau = M.evaluate(O.to_ints(O.FR381, q), [t % R for t in taus[i + 1:]], R) if len(q) > 1 else O.to_ints(O.FR381, q)[0]
            assert pt(proofs[i]) == M.g1_mul(M.G1, q_tau)
            acc = (acc + q_tau * (taus[i] - opening[i])) % R
        assert acc == (f_tau - v) % R
        d = derived[k["src"]]
        assert f_tau == d["f_tau"] % R and v == d["f_open"] % R
        if "commit_x" in d:
            assert pt(commit) == (int(d["commit_x"], 16), int(d["commit_y"], 16))
        assert len(proofs) == n


def test_kzg_length_asserts():
    import pytest
    pts = O.kzg_setup_g1(O.from_ints(O.FR381, [5, 2]))
    with pytest.raises(O.OraclePanic) as e:                # multilinear_kzg.rs:29-33
        O.kzg_commit(O.from_ints(O.FR381, [1, 2]), pts)
    assert e.value.code == O.E_KZG_LEN
    with pytest.raises(O.OraclePanic):                     # :55-59
        O.kzg_open(O.from_ints(O.FR381, [1, 2, 3, 4]), pts, O.from_ints(O.FR381, [7]))
    with pytest.raises(O.OraclePanic):                     # :60-64
        O.kzg_open(O.from_ints(O.FR381, [1, 2, 3, 4]), pts, O.from_ints(O.FR381, [7, 8]), n_g2=3)


def test_commit_with_zero_scalars_and_infinity_bases():
    # tau_i in {0,1} makes Lagrange coefficients 0 -> bases at infinity (MSM exactness)
    pts = O.kzg_setup_g1(O.from_ints(O.FR381, [1, 0, 7]))
    assert sum(1 for p in pts if pt(p) is None) == 6
    vals = O.from_ints(O.FR381, [3, 0, 9, 1, 0, 0, 5, 11])
    f_tau = _eval_at([3, 0, 9, 1, 0, 0, 5, 11], [1, 0, 7])
    assert pt(O.kzg_commit(vals, pts)) == M.g1_mul(M.G1, f_tau)
    assert pt(O.kzg_commit(O.from_ints(O.FR381, [0] * 8), pts)) is None


def test_cpu_pippenger_baseline_equals_naive_commit():
    """the all-cores CPU baseline bench.py times beside the GPU MSM (not reference code) computes the commitment the
    reference's naive sum does (multilinear_kzg.rs:37-42), incl. zero scalars, infinity and repeated bases"""
    rng = random.Random(11)
    g = O.g1_generator()
    n = 37
    pts = np.stack([O.g1_mul_fr(g, O.from_ints(O.FR381, [5 + 7 * i])[0]) for i in range(n)])
    pts[3] = 0                                   # infinity
    pts[9] = pts[8]
    ks = [rng.randrange(R) for _ in range(n)]
    ks[0], ks[1], ks[2] = 0, 1, R - 1
    sc = O.from_ints(O.FR381, ks)
    want = O.g1_affine_ints(O.kzg_commit(sc, pts))
    for c, slices in ((1, 1), (4, 3), (8, 2), (13, 5), (16, 1)):
        assert O.g1_affine_ints(O.msm_pippenger(sc, pts, c, slices)) == want, (c, slices)
    secs, threads, c = O.bench_pippenger_mt(sc, pts)
    assert secs >= 0 and threads >= 1
