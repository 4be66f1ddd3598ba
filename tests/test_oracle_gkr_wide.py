"""CPU: the oracle's linear-time restatement of gkr_protocol::prove for layers of any width (oracle/gkr_wide.c: round polynomials summed gate
by gate from the definition f(b,c) = add(b,c)(W(b) + W(c)) + mul(b,c) W(b) W(c), gkr_protocol.rs:57-143, utils.rs:8-68) against the two
models that exist already: the reference-shaped C restatement on circuits of the reference's shape (oracle/gkr.c), and the generalised
DENSE big-int model on wide shapes (oracle/pymodel.py gkr_prove_wide).  The GPU suite then compares the product's sparse prover with
gkr_wide.c at sizes no dense model can hold (tests/test_gpu_gkr_sparse.py)."""
import random

import numpy as np
import pytest

from oracle import oracle as O
from oracle import pymodel as M

WIDE_SHAPES = [(3, 5, 4, 3), (2, 6, 3, 5, 2), (1, 4, 2), (2, 2, 2), (4, 1, 3), (1, 1, 1), (3, 3), (5, 2), (2, 5), (1, 6, 1), (6, 1, 4),
               (2, 3, 4, 5, 1), (4, 4, 4, 4), (3, 1), (1, 5, 5), (5, 5, 1), (2, 4, 6, 2), (6, 3), (3, 6), (4, 2, 4, 2), (2, 1, 2, 1, 2, 1)]


def ints(field, arr):
    return O.to_ints(field, arr)


@pytest.mark.parametrize("shape", WIDE_SHAPES)
def test_linear_time_oracle_equals_the_dense_wide_model(shape):
    *out_bits, in_last = shape
    widths = list(out_bits) + [in_last]
    for f in (O.FR381, O.BN254_FQ):
        p = O.modulus(f)
        rng = random.Random(hash(shape) % 997 + f)
        spec = []
        for l in range(len(out_bits)):
            n_out, n_in = 1 << widths[l], 1 << widths[l + 1]
            seen = set()
            for _ in range(rng.randrange(n_out // 2 + 1, 2 * n_out + 2)):          # unused outputs, several gates per output
                seen.add((rng.randrange(n_in), rng.randrange(n_in), rng.randrange(n_out), rng.choice([0, 1])))
            spec.append(sorted(seen, key=lambda g: (g[2], g[0], g[1], g[3])))
        xs = [rng.choice([0, 1, p - 1, rng.randrange(p)]) for _ in range(1 << in_last)]
        want = M.gkr_prove_wide(spec, out_bits, xs, p)
        rows = [np.array(layer, np.uint64).reshape(-1, 4) for layer in spec]
        got = O.gkr_prove_wide(f, rows, out_bits, O.from_ints(f, xs))
        assert ints(f, got["circuit_output"]) == want["circuit_output"]
        assert ints(f, got["output_challenges"]) == want["output_challenges"]
        assert ints(f, got["layer_claims"]) == want["layer_claims"]
        assert [ints(f, c) for c in got["coeffs"]] == want["coeffs"]
        assert ints(f, got["challenges"]) == want["challenges"]
        assert ints(f, got["wb_evals"]) == want["wb"] and ints(f, got["wc_evals"]) == want["wc"]
        assert ints(f, got["claimed_sum"].reshape(1, -1)) == [want["claimed_sum"]]


def test_linear_time_oracle_equals_the_reference_shaped_restatement():
    """circuits of the reference's own shape (layer i has 2^i outputs padded to two, gkr_protocol.rs:247-292): the whole proof equals
    oracle/gkr.c's, which follows the reference's dense code line by line"""
    f = O.BN254_FQ
    circuits = [([[(0, 1, 0, 1)], [(0, 1, 0, 0), (2, 3, 1, 1)]], [2, 3, 4, 5]),
                ([[(0, 1, 0, 0)], [(0, 1, 0, 0), (2, 3, 1, 1)], [(0, 1, 0, 0), (2, 3, 1, 1), (4, 5, 2, 1), (6, 7, 3, 1)]], [1, 2, 3, 4, 5, 6, 7, 8])]
    for layers, xs in circuits:
        ref = O.gkr_prove(f, layers, O.from_ints(f, xs))
        out_bits = [max(1, l) for l in range(len(layers))]          # the reference pads its one output to two entries (:39-47)
        rows = [np.array(layer, np.uint64).reshape(-1, 4) for layer in layers]
        got = O.gkr_prove_wide(f, rows, out_bits, O.from_ints(f, xs))
        assert np.array_equal(got["coeffs"], ref["coeffs"]) and np.array_equal(got["challenges"], ref["challenges"])
        assert np.array_equal(got["layer_claims"], ref["layer_claims"]) and np.array_equal(got["claimed_sum"], ref["claimed_sum"])
        assert np.array_equal(got["wb_evals"], ref["wb_evals"]) and np.array_equal(got["wc_evals"], ref["wc_evals"])
