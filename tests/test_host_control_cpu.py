"""CPU checks of the product's host control path (transcript, univariate interpolation): these are
host code by design (O(rounds) work), pinned against the oracle and the golden vectors."""
import random

import numpy as np
import pytest

import __graft_entry__ as G
from oracle import oracle as O


@pytest.fixture(scope="module")
def zk():
    G.build()
    return G.import_package()


def test_keccak_and_transcript_kats(zk, derived_kats):
    for kat in derived_kats["keccak256"]:
        assert zk.sumcheck.keccak256(bytes.fromhex(kat["msg_hex"])).hex() == kat["digest"]
    kat = derived_kats["transcript"]
    t = zk.Transcript.new()
    t.append(kat["append"].encode())
    assert t.sample_random_challenge().hex() == kat["first_sample"]
    assert zk.to_ints(2, t.random_challenge_as_field_element(2)) == [int(kat["then_challenge_bn254_fq"], 16)]


def test_transcript_matches_oracle_on_random_schedules(zk):
    rng = random.Random(11)
    for field in (0, 1, 2, 3):
        a, b = zk.Transcript(), O.Transcript()
        for step in range(40):
            if rng.random() < 0.6:
                data = bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 31, 32, 96, 135, 136, 137, 300, 1000, 4103])))
                a.append(data)
                b.append(data)
            elif rng.random() < 0.5:
                assert a.sample_random_challenge() == b.sample_random_challenge()
            else:
                assert np.array_equal(a.random_challenge_as_field_element(field), b.random_challenge_as_field_element(field))


def test_univariate_helpers(zk, ref_kats):
    for k in ref_kats["lagrange_interpolate"]:
        got = zk.sumcheck.lagrange_interpolate(2, zk.from_ints(2, k["xs"]), zk.from_ints(2, k["ys"]))
        assert zk.to_ints(2, got) == k["expect"]
    for k in ref_kats["univariate_evaluate"]:
        assert zk.to_ints(2, zk.sumcheck.uni_evaluate(2, zk.from_ints(2, k["coeffs"]), zk.from_ints(2, [k["x"]])[0])) == [k["expect"]]
    rng = random.Random(2)
    for field in (0, 2):
        p = O.modulus(field)
        for n in (1, 2, 3, 4):
            xs = list(range(n))
            ys = [rng.randrange(p) for _ in range(n)]
            got = zk.sumcheck.lagrange_interpolate(field, zk.from_ints(field, xs), zk.from_ints(field, ys))
            assert np.array_equal(got, O.lagrange_interpolate(field, O.from_ints(field, xs), O.from_ints(field, ys)))
