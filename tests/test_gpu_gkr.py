"""GPU parity: circuit model and the GKR prover vs the oracle (gkr/src/gkr_protocol.rs:238-300,
circuit/src/arithmetic_circuit.rs:213-385)."""
import random

import numpy as np
import pytest

import __graft_entry__ as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu
FID = {"bn254_fq": 2, "bls12_381_fr": 0}
OPS = {"add": 0, "mul": 1}


@pytest.fixture(scope="module")
def zk():
    zk = G.import_package()
    from zkmle_amd import _lib
    _lib.check(zk.lib().zk_init(0))
    return zk


def mk_circuit(zk, f, spec):
    return zk.Circuit.new(f, [zk.Layer.new([zk.Gate.new(g[0], g[1], g[2], OPS[g[3]] if isinstance(g[3], str) else g[3]) for g in layer]) for layer in spec])


def olayers(spec):
    return [[(g[0], g[1], g[2], OPS[g[3]] if isinstance(g[3], str) else g[3]) for g in layer] for layer in spec]


def test_circuit_reference_kats(zk, ref_kats):
    for k in ref_kats["circuit_evaluate"]:
        f = FID[k["field"]]
        res = mk_circuit(zk, f, k["layers"]).evaluate(zk.from_ints(f, k["inputs"]))
        if "expect_layers" in k:
            assert [zk.to_ints(f, e) for e in res.layer_evaluations] == k["expect_layers"], k["src"]
            assert zk.to_ints(f, res.output) == k["expect_layers"][0]
        else:
            assert zk.to_ints(f, res.output) == k["expect_output"]
    for li, nv in ref_kats["num_of_layer_variables"]["expect"].items():
        assert zk.gkr.num_of_layer_variables(int(li)) == nv
    for k in ref_kats["add_i_mul_i"]:
        spec = [[]] * k["layer_index"] + [k["gates"]]
        c = mk_circuit(zk, 2, spec)
        a, m = c.add_i_and_mul_i_mle(k["layer_index"])
        av, mv = a.to_ints(), m.to_ints()
        assert len(av) == k["len"]
        assert [i for i, v in enumerate(av) if v] == k["add_ones"] and set(av) <= {0, 1}
        assert [i for i, v in enumerate(mv) if v] == k["mul_ones"] and set(mv) <= {0, 1}
    assert zk.gkr.convert_to_binary_and_to_decimal(1, 1, 0, 1) == 17      # "1"+"00"+"01"  :372
    assert zk.gkr.convert_to_binary_and_to_decimal(1, 0, 2, 3) == 11      # "0"+"10"+"11"  :376


def check_against_oracle(zk, f, spec, inputs_ints):
    circuit = mk_circuit(zk, f, spec)
    x = zk.from_ints(f, inputs_ints)
    proof = zk.gkr.prove(circuit, x)
    want = O.gkr_prove(f, olayers(spec), x)
    assert np.array_equal(proof.circuit_output, want["circuit_output"])
    assert np.array_equal(proof.claimed_sum, want["claimed_sum"])
    claims, co, ch = proof._flat
    assert np.array_equal(claims, want["layer_claims"])
    assert np.array_equal(co, want["coeffs"])
    assert np.array_equal(ch, want["challenges"])
    assert np.array_equal(proof.wb_evaluations, want["wb_evals"]) and np.array_equal(proof.wc_evaluations, want["wc_evals"])
    assert zk.gkr.verify(circuit, proof, x) is True
    assert O.gkr_verify(f, olayers(spec), want, x) is True
    bad = zk.from_ints(f, [v + 1 for v in inputs_ints])
    assert zk.gkr.verify(circuit, proof, bad) is False
    return proof


def test_gkr_reference_roundtrips_and_derived(zk, ref_kats, derived_kats):
    for k in ref_kats["gkr_roundtrip"]:                       # gkr_protocol.rs:246-299
        check_against_oracle(zk, FID[k["field"]], k["layers"], k["inputs"])
    for d in derived_kats["gkr_prove"]:
        f = FID[d["field"]]
        proof = check_against_oracle(zk, f, d["layers"], d["inputs"])
        assert zk.to_ints(f, proof.circuit_output) == d["output"]
        assert zk.to_ints(f, proof.claimed_sum) == [int(d["claimed_sum"], 16)]
    k = ref_kats["bench_inputs"][1]                           # gkr/benches/gkr_protocol_benchmark.rs:7-15 (BLS12-381 Fr there)
    check_against_oracle(zk, 0, k["layers"], k["inputs"])


@pytest.mark.parametrize("depth", [1, 2, 3, 4, 5, 6, 7, 8])
def test_gkr_random_circuits(zk, depth):
    """random dense circuits of the reference's shape: layer i has 2^i gates reading 2^(i+1) wires.  Depth 8 is bench.py's `paths.gkr_dense` shape:
    the last layer's add_i / mul_i have 2^23 entries and its alpha / beta combination folds 7 variables in one pass (fold_alpha_beta_kernel)"""
    rng = random.Random(depth)
    f = 0
    p = O.modulus(f)
    spec = []
    for i in range(depth):
        n_out, n_in = (1 << i), (1 << (i + 1))
        layer = [[rng.randrange(n_in), rng.randrange(n_in), o, rng.choice(["add", "mul"])] for o in range(n_out)]
        layer += [[rng.randrange(n_in), rng.randrange(n_in), rng.randrange(n_out), "mul"]]   # a second gate on one output (+=, :96)
        spec.append(layer)
    inputs = [rng.randrange(p) for _ in range(1 << depth)]
    check_against_oracle(zk, f, spec, inputs)


@pytest.mark.parametrize("depth", [1, 3, 5, 8])
def test_gkr_random_circuits_on_the_dense_tables(zk, depth, monkeypatch):
    """zk_gkr_prove proves a well-formed circuit from its gate lists; ZK_GKR_DENSE_TABLES=1 keeps the reference's dense add_i / mul_i representation
    (the path that also serves repeated gates and the shapes the reference panics on): the same proof, against the oracle"""
    monkeypatch.setenv("ZK_GKR_DENSE_TABLES", "1")
    test_gkr_random_circuits(zk, depth)


def test_gkr_both_representations_same_bytes(zk, monkeypatch):
    rng = random.Random(77)
    spec = []
    for i in range(6):
        n_out, n_in = (1 << i), (1 << (i + 1))
        spec.append([[rng.randrange(n_in), rng.randrange(n_in), o, rng.choice(["add", "mul"])] for o in range(n_out)])
    x = zk.from_ints(0, [rng.randrange(O.modulus(0)) for _ in range(1 << 6)])
    circuit = mk_circuit(zk, 0, spec)
    a = zk.gkr.prove(circuit, x)
    a2 = zk.gkr.prove(circuit, x)                              # the compiled gate lists of the first call, reused
    monkeypatch.setenv("ZK_GKR_DENSE_TABLES", "1")
    b = zk.gkr.prove(circuit, x)
    for p in (a2, b):
        assert np.array_equal(a.circuit_output, p.circuit_output) and np.array_equal(a.claimed_sum, p.claimed_sum)
        assert all(np.array_equal(u, v) for u, v in zip(a._flat, p._flat))
        assert np.array_equal(a.wb_evaluations, p.wb_evaluations) and np.array_equal(a.wc_evaluations, p.wc_evaluations)
    # a repeated gate is ONE entry of the dense predicate: such circuits stay with the dense tables whatever the switch says
    monkeypatch.delenv("ZK_GKR_DENSE_TABLES")
    spec[3].append(list(spec[3][0]))
    proof = zk.gkr.prove(mk_circuit(zk, 0, spec), x)
    want = O.gkr_prove(0, olayers(spec), x)                    # (the reference's own verifier may well reject this proof: evaluation counts the gate twice)
    claims, co, ch = proof._flat
    assert np.array_equal(claims, want["layer_claims"]) and np.array_equal(co, want["coeffs"]) and np.array_equal(ch, want["challenges"])
    assert np.array_equal(proof.wb_evaluations, want["wb_evals"]) and np.array_equal(proof.wc_evaluations, want["wc_evals"])


@pytest.mark.parametrize("field", [2, 3])
def test_gkr_random_circuit_other_fields(zk, field):
    """the same against the oracle on BN254 Fq / Fr, depth 6 (the one-pass alpha / beta fold over 1 .. 5 variables in those fields' arithmetic)"""
    rng = random.Random(600 + field)
    p = O.modulus(field)
    spec = []
    for i in range(6):
        n_out, n_in = (1 << i), (1 << (i + 1))
        spec.append([[rng.randrange(n_in), rng.randrange(n_in), o, rng.choice(["add", "mul"])] for o in range(n_out)])
    check_against_oracle(zk, field, spec, [rng.randrange(p) for _ in range(1 << 6)])


def test_gkr_shape_panics(zk):
    spec = [[[0, 1, 0, "mul"]], [[0, 1, 0, "add"], [2, 3, 1, "mul"], [4, 5, 2, "mul"], [6, 7, 3, "add"]]]
    with pytest.raises(zk.ReferencePanic):
        zk.gkr.prove(mk_circuit(zk, 0, spec), zk.from_ints(0, list(range(8))))
