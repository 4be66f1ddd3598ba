"""The two restatements (C oracle, Python big-int model) against each other on random inputs."""
import random

import numpy as np
import pytest

from oracle import oracle as O
from oracle import pymodel as M

FIELDS = [(O.FR381, "bls12_381_fr"), (O.BN254_FQ, "bn254_fq"), (O.FQ381, "bls12_381_fq")]


@pytest.mark.parametrize("fid,name", FIELDS)
def test_fold_every_variable(fid, name):
    rng = random.Random(fid)
    p = M.P[name]
    for n in range(1, 7):
        tab = [rng.randrange(p) for _ in range(1 << n)]
        r = rng.randrange(p)
        for var in range(n):
            got = O.to_ints(fid, O.partial_evaluate(fid, O.from_ints(fid, tab), var, O.from_ints(fid, [r])))
            assert got == M.partial_evaluate(tab, var, r, p)
    with pytest.raises(O.OraclePanic):
        O.partial_evaluate(fid, O.from_ints(fid, [1, 2, 3, 4]), 2, O.from_ints(fid, [5]))


@pytest.mark.parametrize("fid,name", FIELDS[:2])
def test_basic_and_gkr_sumcheck_random(fid, name):
    rng = random.Random(100 + fid)
    p = M.P[name]
    for n in (1, 2, 5):
        tab = [rng.randrange(p) for _ in range(1 << n)]
        cs, rp, ch = O.sumcheck_basic_prove(fid, O.from_ints(fid, tab))
        mcs, mrp, mch = M.sumcheck_basic_prove(tab, p)
        assert O.to_ints(fid, cs) == [mcs] and O.to_ints(fid, ch) == mch
        assert [O.to_ints(fid, r) for r in rp] == mrp
        assert O.sumcheck_basic_verify(fid, O.from_ints(fid, tab), cs, rp)
    for nprod, nfac, n in ((2, 2, 3), (3, 2, 2), (2, 3, 3)):
        tabs = [[[rng.randrange(p) for _ in range(1 << n)] for _ in range(nfac)] for _ in range(nprod)]
        arr = np.stack([np.stack([O.from_ints(fid, t) for t in prod]) for prod in tabs])
        claimed = sum(O.to_ints(fid, O.sumpoly_reduce(fid, arr))) % p
        co, ch = O.sumcheck_gkr_prove(fid, arr, O.from_ints(fid, [claimed]), O.Transcript())
        mco, mch = M.sumcheck_gkr_prove(tabs, claimed, M.Transcript(), p)
        assert [O.to_ints(fid, r) for r in co] == mco and O.to_ints(fid, ch) == mch
        ok, vch, last = O.sumcheck_gkr_verify(fid, O.from_ints(fid, [claimed]), co, O.Transcript())
        assert ok and O.to_ints(fid, last) == O.to_ints(fid, O.sumpoly_evaluate(fid, arr, ch))


def test_gkr_random_inputs():
    rng = random.Random(42)
    fid, name = O.FR381, "bls12_381_fr"
    p = M.P[name]
    # the reference ties width to depth: layer i reads a 2^(i+1)-entry layer (arithmetic_circuit.rs:166-178)
    layers = [[(0, 1, 0, O.MUL), (1, 0, 1, O.ADD)],
              [(0, 3, 0, O.ADD), (2, 1, 1, O.MUL)],
              [(0, 1, 0, O.MUL), (2, 3, 1, O.ADD), (7, 5, 2, O.MUL), (6, 4, 3, O.ADD), (0, 7, 3, O.MUL)]]
    inputs = [rng.randrange(p) for _ in range(8)]
    proof = O.gkr_prove(fid, layers, O.from_ints(fid, inputs))
    mp = M.gkr_prove(layers, inputs, p)
    assert O.to_ints(fid, proof["claimed_sum"]) == [mp["claimed_sum"]]
    assert O.to_ints(fid, proof["circuit_output"]) == mp["circuit_output"]
    assert O.gkr_verify(fid, layers, proof, O.from_ints(fid, inputs))
    assert M.gkr_verify(layers, mp, inputs, p)
    # a layer whose width does not match its depth panics in ProductPolynomial::new
    bad = [[(0, 1, 0, O.MUL)], [(0, 1, 0, O.ADD), (2, 3, 1, O.MUL), (4, 5, 2, O.MUL), (6, 7, 3, O.ADD)]]
    with pytest.raises(O.OraclePanic):
        O.gkr_prove(fid, bad, O.from_ints(fid, inputs))


def test_kzg_random():
    rng = random.Random(8)
    taus = [rng.randrange(M.R) for _ in range(3)]
    vals = [rng.randrange(M.R) for _ in range(8)]
    opening = [rng.randrange(M.R) for _ in range(3)]
    pts = O.kzg_setup_g1(O.from_ints(O.FR381, taus))
    mpts = M.kzg_setup_g1(taus)
    assert O.g1_affine_ints(O.kzg_commit(O.from_ints(O.FR381, vals), pts)) == M.kzg_commit(vals, mpts)
    ev, proofs = O.kzg_open(O.from_ints(O.FR381, vals), pts, O.from_ints(O.FR381, opening))
    mv, mproofs = M.kzg_open(vals, mpts, opening)
    assert O.to_ints(O.FR381, ev) == [mv] and [O.g1_affine_ints(q) for q in proofs] == mproofs
