"""GPU: the sparse (linear-time) GKR prover.  On circuits of the reference's shape its proof must be
bit-identical to the dense definition (oracle = restatement of gkr/src/gkr_protocol.rs); on wide circuits
(which the reference cannot express) it is checked by the verifier equations with an independently
evaluated wiring predicate."""
import random

import numpy as np
import pytest

import __graft_entry__ as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu
OPS = {"add": 0, "mul": 1}


@pytest.fixture(scope="module")
def zk():
    zk = G.import_package()
    from zkmle_amd import _lib
    _lib.check(zk.lib().zk_init(0))
    return zk


def rows_of(spec):
    return [np.array([[g[0], g[1], g[2], OPS[g[3]] if isinstance(g[3], str) else g[3]] for g in layer], np.uint64) for layer in spec]


def ref_shape_bits(nlayers):
    return [1] + list(range(1, nlayers))          # layer 0: one a-bit (arithmetic_circuit.rs:167-169), layer i: i bits


def check_dense_parity(zk, f, spec, inputs_ints):
    x = zk.from_ints(f, inputs_ints)
    ob = ref_shape_bits(len(spec))
    proof = zk.gkr.sparse_prove(f, rows_of(spec), ob, x)
    want = O.gkr_prove(f, [[(g[0], g[1], g[2], OPS[g[3]] if isinstance(g[3], str) else g[3]) for g in layer] for layer in spec], x)
    out = want["circuit_output"]
    assert np.array_equal(proof.circuit_output[: len(out)], out) and not proof.circuit_output[len(out):].any()
    assert np.array_equal(proof.claimed_sum, want["claimed_sum"])
    assert np.array_equal(proof.layer_claims, want["layer_claims"])
    assert np.array_equal(proof.coeffs, want["coeffs"])
    assert np.array_equal(proof.challenges, want["challenges"])
    assert np.array_equal(proof.wb_evals, want["wb_evals"]) and np.array_equal(proof.wc_evals, want["wc_evals"])
    assert zk.gkr.sparse_verify(f, rows_of(spec), ob, proof, x) is True
    return proof


def test_reference_circuits_bit_identical_to_dense(zk, ref_kats, derived_kats):
    for k in ref_kats["gkr_roundtrip"]:                       # gkr_protocol.rs:246-299
        check_dense_parity(zk, 2, k["layers"], k["inputs"])
    for d in derived_kats["gkr_prove"]:
        p = check_dense_parity(zk, 2, d["layers"], d["inputs"])
        assert zk.to_ints(2, p.claimed_sum) == [int(d["claimed_sum"], 16)]


@pytest.mark.parametrize("depth", [1, 2, 3, 4, 5])
def test_random_reference_shaped_circuits(zk, depth):
    rng = random.Random(40 + depth)
    f = 0
    p = O.modulus(f)
    spec = []
    for i in range(depth):
        n_out, n_in = (1 << i), (1 << (i + 1))
        seen, layer = set(), []
        for o in range(n_out):
            for _ in range(rng.choice([1, 1, 2])):           # some outputs are sums of two gates (+=, :96)
                g = (rng.randrange(n_in), rng.randrange(n_in), o, rng.choice(["add", "mul"]))
                if g not in seen:
                    seen.add(g)
                    layer.append(list(g))
        spec.append(layer)
    check_dense_parity(zk, f, spec, [rng.randrange(p) for _ in range(1 << depth)])


@pytest.mark.parametrize("bits", [(3, 5, 4, 6), (10, 10, 10, 10), (1, 12, 3, 7)])
def test_wide_circuits_verify(zk, bits):
    """layers of arbitrary widths and random wiring (not expressible in the reference): prove, then verify"""
    rng = np.random.default_rng(sum(bits))
    f = 0
    *out_bits, in_bits_last = bits
    widths = list(out_bits) + [in_bits_last]
    rows = []
    for l in range(len(out_bits)):
        n_out, n_in = 1 << widths[l], 1 << widths[l + 1]
        g = np.zeros((n_out, 4), np.uint64)
        g[:, 0] = rng.integers(0, n_in, n_out)
        g[:, 1] = rng.integers(0, n_in, n_out)
        g[:, 2] = np.arange(n_out)
        g[:, 3] = rng.integers(0, 2, n_out)
        rows.append(g)
    x = zk.MultilinearPolynomial.random(f, 1 << in_bits_last, 5).evaluated_values
    proof = zk.gkr.sparse_prove(f, rows, out_bits, x)
    assert zk.gkr.sparse_verify(f, rows, out_bits, proof, x) is True
    # the circuit evaluation agrees with the reference's evaluator (oracle) on the same gate lists
    evs = O.circuit_evaluate(f, [[tuple(int(v) for v in r) for r in layer] for layer in rows], x)
    assert np.array_equal(proof.circuit_output[: len(evs[0])], evs[0])
    bad = x.copy()
    bad[3, 0] ^= np.uint64(1)
    assert zk.gkr.sparse_verify(f, rows, out_bits, proof, bad) is False
    tampered = zk.gkr.SparseProof(**{**proof.__dict__, "wb_evals": proof.wb_evals.copy()})
    if len(tampered.wb_evals):
        tampered.wb_evals[0, 0] ^= np.uint64(1)
        assert zk.gkr.sparse_verify(f, rows, out_bits, tampered, x) is False


@pytest.mark.gpu
@pytest.mark.parametrize("wiring,lg", [("random", 18), ("regular", 18), ("random", 22), ("regular", 22)])
def test_config4_prove_verify(zk, wiring, lg):
    """BASELINE config 4 (depth 3, 2^22 gates per layer, both wirings of SURVEY 8d; 2^18 as a quicker case), beyond what the
    dense oracle can hold: the proof passes the sparse verifier (sumcheck equations + the wiring predicate evaluated
    independently on the device), the compiled-circuit path gives the same proof, and a flipped coefficient is rejected."""
    field, depth = 0, 3
    n = 1 << lg
    rng = np.random.default_rng(0x5EED0004)
    rows, out_bits = [], []
    for _ in range(depth):
        g = np.zeros((n, 4), np.uint64)
        if wiring == "random":
            g[:, 0] = rng.integers(0, n, n); g[:, 1] = rng.integers(0, n, n)
        else:
            g[:, 0] = (2 * np.arange(n)) % n; g[:, 1] = (2 * np.arange(n) + 1) % n
        g[:, 2] = np.arange(n); g[:, 3] = rng.integers(0, 2, n)
        rows.append(g); out_bits.append(lg)
    x = zk.MultilinearPolynomial.random(field, n, 0x5EED0004).evaluated_values
    proof = zk.gkr.sparse_prove(field, rows, out_bits, x)
    assert zk.gkr.sparse_verify(field, rows, out_bits, proof, x)
    circuit = zk.gkr.SparseCircuit(rows, out_bits, n)
    again = zk.gkr.sparse_prove(field, None, None, x, circuit=circuit)
    assert np.array_equal(again.coeffs, proof.coeffs) and np.array_equal(again.challenges, proof.challenges)
    proof.coeffs[5, 1, 0] ^= np.uint64(1)
    assert not zk.gkr.sparse_verify(field, rows, out_bits, proof, x)


WIDE_SHAPES = [  # (out_bits per layer ..., log2 #inputs): none of them expressible in the reference (width != depth, k0 > 1 ...)
    (3, 5, 4, 3), (2, 6, 3, 5, 2), (1, 4, 2), (2, 2, 2), (4, 1, 3), (1, 1, 1), (3, 3), (5, 2), (2, 5), (1, 6, 1), (6, 1, 4),
    (2, 3, 4, 5, 1), (4, 4, 4, 4), (3, 1), (1, 5, 5), (5, 5, 1), (2, 4, 6, 2), (6, 3), (3, 6), (1, 2, 3, 4, 5, 2), (4, 2, 4, 2), (5, 3, 5),
    (2, 1, 2, 1, 2, 1), (6, 6, 2),
]


@pytest.mark.parametrize("shape", WIDE_SHAPES)
def test_wide_circuits_bit_identical_to_generalised_dense_model(zk, shape):
    """Independent oracle for circuits the reference cannot express: oracle/pymodel.py generalises the DENSE definition
    (0/1 wiring tables indexed a||b||c, alpha/beta folding, dense f(b,c), arithmetic_circuit.rs:126-200, utils.rs:8-68,
    gkr_protocol.rs:57-143) to per-layer widths with plain big-int loops -- no gate lists, no eq tables, no two-phase split.
    The sparse prover's whole proof (every coefficient, challenge, layer claim, wb / wc, output challenges) must equal it."""
    from oracle import pymodel as M
    *out_bits, in_last = shape
    widths = list(out_bits) + [in_last]
    for f in (0, 2):
        p = O.modulus(f)
        rng = random.Random(hash(shape) % 1000 + f)
        spec = []
        for l in range(len(out_bits)):
            n_out, n_in = 1 << widths[l], 1 << widths[l + 1]
            seen = set()
            for _ in range(rng.randrange(n_out // 2 + 1, 2 * n_out + 2)):      # some outputs unused, some fed by several gates (+=, :96)
                seen.add((rng.randrange(n_in), rng.randrange(n_in), rng.randrange(n_out), rng.choice([0, 1])))
            spec.append(sorted(seen, key=lambda g: (g[2], g[0], g[1], g[3])))
        xs = [rng.choice([0, 1, p - 1, rng.randrange(p)]) for _ in range(1 << in_last)]
        want = M.gkr_prove_wide(spec, out_bits, xs, p)
        rows = [np.array(layer, np.uint64).reshape(-1, 4) for layer in spec]
        x = zk.from_ints(f, xs)
        proof = zk.gkr.sparse_prove(f, rows, out_bits, x)
        assert zk.to_ints(f, proof.circuit_output) == want["circuit_output"]
        assert zk.to_ints(f, proof.output_challenges) == want["output_challenges"]
        assert zk.to_ints(f, proof.layer_claims) == want["layer_claims"]
        assert [zk.to_ints(f, c) for c in proof.coeffs] == want["coeffs"]
        assert zk.to_ints(f, proof.challenges) == want["challenges"]
        assert zk.to_ints(f, proof.wb_evals) == want["wb"] and zk.to_ints(f, proof.wc_evals) == want["wc"]
        assert zk.to_ints(f, proof.claimed_sum.reshape(1, -1)) == [want["claimed_sum"]]
        assert zk.gkr.sparse_verify(f, rows, out_bits, proof, x) is True


def test_skewed_wiring_bit_identical_to_generalised_dense_model(zk):
    """Most gates of a layer share ONE left (resp. right) index: the grouped gate list has runs far longer than the 256 gates a pass of
    the gate-parallel table kernels stages (csrc/zkmle_gkr_sparse.hip grouped_pair_sums), next to many empty groups."""
    from oracle import pymodel as M
    f, out_bits, in_last = 0, [6, 6], 6
    p = O.modulus(f)
    rng = random.Random(77)
    spec = []
    for l in range(2):
        seen = set()
        hot = rng.randrange(64)
        while len(seen) < 700:                                  # 700 gates under one left index
            seen.add((hot, rng.randrange(64), rng.randrange(64), rng.choice([0, 1])))
        hot_r = rng.randrange(64)
        while len(seen) < 1300:                                 # 600 more under one right index
            seen.add((rng.randrange(64), hot_r, rng.randrange(64), rng.choice([0, 1])))
        for _ in range(60):
            seen.add((rng.randrange(64), rng.randrange(64), rng.randrange(64), rng.choice([0, 1])))
        spec.append(sorted(seen, key=lambda g: (g[2], g[0], g[1], g[3])))
    xs = [rng.randrange(p) for _ in range(1 << in_last)]
    want = M.gkr_prove_wide(spec, out_bits, xs, p)
    rows = [np.array(layer, np.uint64).reshape(-1, 4) for layer in spec]
    x = zk.from_ints(f, xs)
    proof = zk.gkr.sparse_prove(f, rows, out_bits, x)
    assert zk.to_ints(f, proof.circuit_output) == want["circuit_output"]
    assert [zk.to_ints(f, c) for c in proof.coeffs] == want["coeffs"]
    assert zk.to_ints(f, proof.challenges) == want["challenges"]
    assert zk.to_ints(f, proof.wb_evals) == want["wb"] and zk.to_ints(f, proof.wc_evals) == want["wc"]
    assert zk.gkr.sparse_verify(f, rows, out_bits, proof, x) is True


def _layers(wiring, lg, depth, seed):
    n = 1 << lg
    rng = np.random.default_rng(seed)
    rows = []
    for _ in range(depth):
        g = np.zeros((n, 4), np.uint64)
        if wiring == "random":
            g[:, 0] = rng.integers(0, n, n); g[:, 1] = rng.integers(0, n, n)
        elif wiring == "regular":
            g[:, 0] = (2 * np.arange(n)) % n; g[:, 1] = (2 * np.arange(n) + 1) % n
        else:                                                   # skewed: half of the gates read ONE left wire, a quarter ONE right wire
            g[:, 0] = rng.integers(0, n, n); g[:, 1] = rng.integers(0, n, n)
            g[: n // 2, 0] = int(rng.integers(0, n)); g[n // 2: 3 * n // 4, 1] = int(rng.integers(0, n))
        g[:, 2] = rng.permutation(n) if wiring == "skewed" else np.arange(n)
        g[:, 3] = rng.integers(0, 2, n)
        if wiring != "regular":                                 # some outputs fed by several gates, some by none (+=, arithmetic_circuit.rs:96)
            g[: n // 16, 2] = g[n // 16: n // 8, 2]
        rows.append(g)
    return rows


@pytest.mark.gpu
@pytest.mark.parametrize("field,wiring,lg", [(0, "random", 10), (0, "regular", 10), (0, "skewed", 10), (2, "random", 10), (2, "skewed", 10),
                                             (0, "random", 13), (0, "regular", 13), (0, "skewed", 13), (2, "random", 13), (2, "regular", 13),
                                             (0, "random", 16), (0, "regular", 16), (0, "skewed", 16), (2, "random", 16),
                                             (0, "random", 18), (2, "skewed", 18)])
def test_mid_size_proofs_bit_identical_to_the_linear_time_oracle(zk, field, wiring, lg):
    """The regime between the dense models (<= 2^6 wires) and config 4 (2^22), where every multi-workgroup path of the sparse prover runs --
    multi-pass table kernels, eq outer products, grid-wide fused rounds with the exchange in their last workgroup, split rounds, tails -- against
    an oracle that shares none of that: oracle/gkr_wide.c sums every round polynomial gate by gate from the definition of f(b, c)
    (gkr_protocol.rs:57-143, utils.rs:8-68; pinned on small shapes by the dense models, tests/test_oracle_gkr_wide.py).  Depth 3, 2^lg
    gates per layer; the WHOLE proof: every coefficient, challenge, layer claim, wb / wc, the output challenges."""
    depth, n = 3, 1 << lg
    rows = _layers(wiring, lg, depth, 0x5EED0400 + lg + 7 * field)
    out_bits = [lg] * depth
    x = zk.MultilinearPolynomial.random(field, n, 0x5EED0004 + lg).evaluated_values
    x[:4] = zk.from_ints(field, [0, 1, O.modulus(field) - 1, 2])
    want = O.gkr_prove_wide(field, rows, out_bits, x)
    proof = zk.gkr.sparse_prove(field, rows, out_bits, x)
    assert np.array_equal(proof.circuit_output, want["circuit_output"])
    assert np.array_equal(proof.output_challenges, want["output_challenges"])
    assert np.array_equal(proof.layer_claims, want["layer_claims"])
    assert np.array_equal(proof.coeffs, want["coeffs"])
    assert np.array_equal(proof.challenges, want["challenges"])
    assert np.array_equal(proof.wb_evals, want["wb_evals"]) and np.array_equal(proof.wc_evals, want["wc_evals"])
    assert np.array_equal(np.asarray(proof.claimed_sum).reshape(-1), want["claimed_sum"])
    circuit = zk.gkr.SparseCircuit(rows, out_bits, n)           # the compiled-circuit path: the same bytes
    again = zk.gkr.sparse_prove(field, None, None, x, circuit=circuit)
    assert np.array_equal(again.coeffs, want["coeffs"]) and np.array_equal(again.challenges, want["challenges"])
