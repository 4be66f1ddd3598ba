"""Randomised differential check of the HIP path against the CPU oracle (test infrastructure, like tests/).

    python tools/fuzz_parity.py --seconds 120 --seed 1

Every iteration draws a field, a size, an operation and a value pattern -- uniformly random elements, or tables made of the
extreme values {0, 1, 2, p-2, p-1}, which maximise the lazily reduced sums of the round kernels -- and compares the result of the
C-ABI call bit for bit with the oracle's.  Prints one JSON line per operation kind with the number of cases run."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G                     # noqa: E402
from oracle import oracle as O                  # noqa: E402

PANICS = {"both_panic": 0, "oracle_panic_sparse_skipped": 0}
FIELDS = [0, 2, 3]                              # BLS12-381 Fr, BN254 Fq, BN254 Fr (the 4-limb fields of the provers)


def table(zk, rng, field, n, mode):
    if mode == "random":
        t = np.zeros((n, zk.limbs(field)), np.uint64)
        assert zk.lib().zk_host_fill_random(field, int(rng.integers(1, 1 << 31)), 0, n, t.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
        return t
    p = O.modulus(field)
    pool = [0, 1, 2, p - 2, p - 1]
    if mode == "max":
        pool = [p - 1]
    vals = [pool[int(i)] for i in rng.integers(0, len(pool), n)]
    return zk.from_ints(field, vals)


def mk_sum(zk, field, tabs):
    S = zk.sumcheck
    return S.SumPolynomial([S.ProductPolynomial([zk.MultilinearPolynomial(field, t) for t in prod]) for prod in tabs])


def op_fold(zk, rng, field, mode):
    logn = int(rng.integers(1, 15))
    t = table(zk, rng, field, 1 << logn, mode)
    var = int(rng.integers(0, logn))
    r = table(zk, rng, field, 1, mode)[0]
    got = zk.MultilinearPolynomial.partial_evaluate(zk.MultilinearPolynomial(field, t), var, r).evaluated_values
    assert np.array_equal(got, O.partial_evaluate(field, t, var, r)), ("fold", field, logn, var, mode)


def op_evaluate(zk, rng, field, mode):
    logn = int(rng.integers(0, 15))
    t = table(zk, rng, field, 1 << logn, mode)
    k = int(rng.integers(0, logn + 1))
    pt = table(zk, rng, field, max(k, 1), mode)[:k]
    got = zk.MultilinearPolynomial(field, t).evaluate(pt)
    assert np.array_equal(np.asarray(got).reshape(-1), np.asarray(O.evaluate(field, t, pt)).reshape(-1)), ("evaluate", field, logn, k, mode)


def op_basic(zk, rng, field, mode, big=False):
    logn = int(rng.integers(17, 23)) if big else int(rng.integers(0, 17))
    t = table(zk, rng, field, 1 << logn, mode)
    prover = zk.Prover.init(field, t)
    proof = prover.prove()
    cs, rp, ch = O.sumcheck_basic_prove(field, t)
    assert np.array_equal(proof.initial_claimed_sum, cs), ("basic claimed", field, logn, mode)
    assert np.array_equal(proof.round_univariate_polynomials.reshape(rp.shape), rp), ("basic rounds", field, logn, mode)
    assert np.array_equal(np.asarray(prover.challenges).reshape(ch.shape), ch), ("basic challenges", field, logn, mode)
    assert zk.Verifier.init().verify(proof) is True


def op_gkr_sumcheck(zk, rng, field, mode, big=False):
    nprod, nfac = int(rng.integers(2, 5)), int(rng.integers(2, 4))
    logn = int(rng.integers(1, 15 if nprod * nfac <= 4 else 13))
    if big:                 # the large-round path: evaluation at 1 derived from the running claim, multi-block reductions
        nprod, nfac, logn = 2, 2, int(rng.integers(16, 19))
    n = 1 << logn
    tabs = np.stack([np.stack([table(zk, rng, field, n, mode) for _ in range(nfac)]) for _ in range(nprod)])
    sp = mk_sum(zk, field, tabs)
    assert np.array_equal(zk.sumcheck.generate_round_univariate(sp), O.gkr_round_univariate(field, tabs)), ("round", field, nprod, nfac, logn, mode)
    claimed = O.vec_sum(field, O.sumpoly_reduce(field, tabs))
    prefix = bytes(rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8))      # any sponge fill
    t_gpu, t_cpu = zk.Transcript(), O.Transcript()
    t_gpu.append(prefix)
    t_cpu.append(prefix)
    res = zk.sumcheck.prove(sp, claimed, t_gpu)
    co, ch = O.sumcheck_gkr_prove(field, tabs, claimed, t_cpu)
    assert np.array_equal(res.round_univariate_polynomials, co), ("gkr coeffs", field, nprod, nfac, logn, mode, len(prefix))
    assert np.array_equal(res.random_challenges, ch), ("gkr challenges", field, nprod, nfac, logn, mode, len(prefix))
    assert t_gpu.sample_random_challenge() == t_cpu.sample_random_challenge()


def op_msm(zk, rng, field, mode):
    n = int(rng.integers(1, 700))
    sc = table(zk, rng, 0, n, mode)
    a = zk.from_ints(0, [int(rng.integers(1, 1 << 62))])[0]
    d = zk.from_ints(0, [int(rng.integers(0, 3)) if mode != "random" else int(rng.integers(1, 1 << 62))])[0]   # d = 0: all bases equal
    bases = zk.G1Bases.synthetic(n, a, d)
    c = int(rng.choice([0, 2, 4, 7, 11, 13, 16, 17, 19, 20, 22]))            # 17 and up: the wide-window sort (round 3)
    got, _ = zk.kzg.msm(zk.MultilinearPolynomial.vector(0, sc), bases, c, True)
    want = O.kzg_commit(sc, bases.points())
    assert np.array_equal(np.asarray(got), np.asarray(want)), ("msm", n, c, mode)
    if int(rng.integers(0, 4)) == 0:                                          # round 3: precomputed window-shifted bases, one bucket set
        used = bases.precompute(int(rng.choice([0, 9, 13, 16, 20])))
        got, st = zk.kzg.msm(zk.MultilinearPolynomial.vector(0, sc), bases, 0, True)
        assert st["window_bits"] == used and np.array_equal(np.asarray(got), np.asarray(want)), ("msm precomputed", n, used, mode)


def op_kzg(zk, rng, field, mode):
    nv = int(rng.integers(1, 8))
    taus = table(zk, rng, 0, nv, "random")
    vals = table(zk, rng, 0, 1 << nv, mode)
    opening = table(zk, rng, 0, nv, mode)
    setup = zk.kzg.TrustedSetup.initialize_setup(taus)
    pts = setup.g1_powers_of_tau.points()
    assert np.array_equal(pts, O.kzg_setup_g1(taus)), ("setup", nv)
    kzg = zk.kzg.MultilinearKZG
    com = kzg.commit_to_polynomial(zk.MultilinearPolynomial(0, vals), setup)
    assert np.array_equal(np.asarray(com), np.asarray(O.kzg_commit(vals, pts))), ("commit", nv, mode)
    proof = kzg.open_and_prove(zk.MultilinearPolynomial(0, vals), setup, opening)
    ev, qs = O.kzg_open(vals, pts, opening)
    assert np.array_equal(np.asarray(proof.evaluation).reshape(-1), np.asarray(ev).reshape(-1)), ("open value", nv, mode)
    assert np.array_equal(np.asarray(proof.proofs), np.asarray(qs)), ("open proofs", nv, mode)


def _random_circuit(rng, depth, dup_ok):
    """the reference's shape: layer i has 2^i outputs reading 2^(i+1) wires; some outputs are sums of several gates (+=,
    arithmetic_circuit.rs:96), some have none"""
    spec = []
    for i in range(depth):
        n_out, n_in = 1 << i, 1 << (i + 1)
        seen, layer = set(), []
        for o in range(n_out):
            for _ in range(int(rng.choice([0, 1, 1, 1, 1, 1, 2, 3]))):
                g = (int(rng.integers(0, n_in)), int(rng.integers(0, n_in)), o, int(rng.integers(0, 2)))
                if dup_ok or g not in seen:
                    seen.add(g)
                    layer.append(g)
        if not layer:
            layer.append((0, min(1, n_in - 1), 0, 0))
        spec.append(layer)
    return spec


def _ints(zk, rng, field, n, mode):
    return table(zk, rng, field, n, mode)


def op_gkr_dense(zk, rng, field, mode):
    depth = int(rng.integers(1, 6))
    spec = _random_circuit(rng, depth, False)
    x = _ints(zk, rng, field, 1 << depth, mode)
    circuit = zk.Circuit.new(field, [zk.Layer.new([zk.Gate.new(*g) for g in layer]) for layer in spec])
    try:
        want = O.gkr_prove(field, spec, x)
    except O.OraclePanic:                       # e.g. a layer whose evaluation vector is not a power of two
        try:
            zk.gkr.prove(circuit, x)
        except zk.ReferencePanic:
            PANICS["both_panic"] += 1
            return
        raise AssertionError(("gkr dense: oracle panics, library does not", field, depth, spec))
    proof = zk.gkr.prove(circuit, x)
    claims, co, ch = proof._flat
    for got, key in ((proof.circuit_output, "circuit_output"), (proof.claimed_sum, "claimed_sum"), (claims, "layer_claims"), (co, "coeffs"),
                     (ch, "challenges"), (proof.wb_evaluations, "wb_evals"), (proof.wc_evaluations, "wc_evals")):
        assert np.array_equal(got, want[key]), ("gkr dense", key, field, depth, mode, spec)
    assert zk.gkr.verify(circuit, proof, x) is True


def op_gkr_sparse(zk, rng, field, mode):
    depth = int(rng.integers(1, 6))
    spec = _random_circuit(rng, depth, False)
    x = _ints(zk, rng, field, 1 << depth, mode)
    rows = [np.array(layer, np.uint64).reshape(-1, 4) for layer in spec]
    ob = [1] + list(range(1, depth))
    try:
        want = O.gkr_prove(field, spec, x)
    except O.OraclePanic:                       # the sparse prover pads where the dense reference panics: nothing to compare
        PANICS["oracle_panic_sparse_skipped"] += 1
        return
    proof = zk.gkr.sparse_prove(field, rows, ob, x)
    out = want["circuit_output"]
    assert np.array_equal(proof.circuit_output[: len(out)], out) and not proof.circuit_output[len(out):].any(), ("sparse out", field, depth, mode, spec)
    for got, key in ((proof.claimed_sum, "claimed_sum"), (proof.layer_claims, "layer_claims"), (proof.coeffs, "coeffs"),
                     (proof.challenges, "challenges"), (proof.wb_evals, "wb_evals"), (proof.wc_evals, "wc_evals")):
        assert np.array_equal(got, want[key]), ("gkr sparse", key, field, depth, mode, spec)
    assert zk.gkr.sparse_verify(field, rows, ob, proof, x) is True


def op_gkr_cf(zk, rng, field, mode, big=False):
    """round 2: two-factor products whose second factor is a constant (zk_sumcheck_gkr_rounds_cf) against the oracle's proof of the
    same SumPolynomial with the constant tables written out"""
    nprod = int(rng.integers(2, 5))
    logn = int(rng.integers(16, 19)) if big else int(rng.integers(1, 15))
    if big:
        nprod = 2
    n = 1 << logn
    MP = zk.MultilinearPolynomial
    cvals = table(zk, rng, field, nprod, mode)
    is_c = [bool(rng.integers(0, 2)) for _ in range(nprod)]
    full = np.stack([np.stack([table(zk, rng, field, n, mode) for _ in range(2)]) for _ in range(nprod)])
    for p in range(nprod):
        if is_c[p]:
            full[p, 1] = cvals[p]
    tables = [(MP(field, full[p, 0]), None if is_c[p] else MP(field, full[p, 1])) for p in range(nprod)]
    claimed = O.vec_sum(field, O.sumpoly_reduce(field, full))
    prefix = bytes(rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8))
    t_gpu, t_cpu = zk.Transcript(), O.Transcript()
    t_gpu.append(prefix)
    t_cpu.append(prefix)
    t_gpu.append(O.fe_to_bytes_be(field, claimed))
    co, ch, fin = zk.sumcheck.gkr_rounds_const_factors(field, tables, cvals, t_gpu)
    wco, wch = O.sumcheck_gkr_prove(field, full, claimed, t_cpu)
    assert np.array_equal(co, wco) and np.array_equal(ch, wch), ("gkr cf", field, nprod, logn, mode, is_c, len(prefix))
    assert t_gpu.sample_random_challenge() == t_cpu.sample_random_challenge()
    assert np.array_equal(fin[0], O.evaluate(field, full[0, 0], wch)), ("gkr cf final", field, nprod, logn, mode)


def op_big_gkr_cf(zk, rng, field, mode):
    op_gkr_cf(zk, rng, field, mode, True)


def op_gkr_wide(zk, rng, field, mode):
    """round 2: sparse GKR on circuits of arbitrary widths against the generalised dense big-int model (oracle/pymodel.py)"""
    from oracle import pymodel as M
    nl = int(rng.integers(1, 5))
    widths = [int(rng.integers(1, 5)) for _ in range(nl + 1)]
    p = O.modulus(field)
    spec = []
    for l in range(nl):
        n_out, n_in = 1 << widths[l], 1 << widths[l + 1]
        seen = set()
        for _ in range(int(rng.integers(1, 2 * n_out + 2))):
            seen.add((int(rng.integers(0, n_in)), int(rng.integers(0, n_in)), int(rng.integers(0, n_out)), int(rng.integers(0, 2))))
        spec.append(sorted(seen, key=lambda g: (g[2], g[0], g[1], g[3])))
    x = _ints(zk, rng, field, 1 << widths[nl], mode)
    xs = zk.to_ints(field, x)
    want = M.gkr_prove_wide(spec, widths[:nl], xs, p)
    rows = [np.array(layer, np.uint64).reshape(-1, 4) for layer in spec]
    proof = zk.gkr.sparse_prove(field, rows, widths[:nl], x)
    assert zk.to_ints(field, proof.circuit_output) == want["circuit_output"], ("wide out", field, widths, spec)
    assert [zk.to_ints(field, c) for c in proof.coeffs] == want["coeffs"], ("wide coeffs", field, widths, mode, spec)
    assert zk.to_ints(field, proof.challenges) == want["challenges"], ("wide challenges", field, widths, mode, spec)
    assert zk.to_ints(field, proof.wb_evals) == want["wb"] and zk.to_ints(field, proof.wc_evals) == want["wc"], ("wide wb/wc", field, widths, spec)
    assert zk.to_ints(field, proof.claimed_sum.reshape(1, -1)) == [want["claimed_sum"]], ("wide claim", field, widths, spec)


def op_elementwise(zk, rng, field, mode):
    MP = zk.MultilinearPolynomial
    logn = int(rng.integers(0, 8))
    a, b = table(zk, rng, field, 1 << logn, mode), table(zk, rng, field, 1 << logn, mode)
    s = table(zk, rng, field, 1, mode)[0]
    assert np.array_equal(MP(field, a).scalar_mul(s).evaluated_values, O.scalar_mul(field, a, s)), ("scalar_mul", field, logn, mode)
    assert np.array_equal(MP.add_polynomials(MP(field, a), MP(field, b)).evaluated_values, O.add_polynomials(field, a, b)), ("add", field, logn, mode)
    assert np.array_equal(MP.polynomial_tensor_add(MP(field, a), MP(field, b)).evaluated_values, O.polynomial_tensor_add(field, a, b))
    assert np.array_equal(MP.polynomial_tensor_mul(MP(field, a), MP(field, b)).evaluated_values, O.polynomial_tensor_mul(field, a, b))
    assert MP(field, a).convert_to_bytes() == O.mle_to_bytes(field, a)


def op_big_gkr(zk, rng, field, mode):
    op_gkr_sumcheck(zk, rng, field, mode, True)


def op_big_basic(zk, rng, field, mode):
    op_basic(zk, rng, field, mode, True)


OPS = {"big_gkr": op_big_gkr, "big_basic": op_big_basic, "big_gkr_cf": op_big_gkr_cf, "gkr_cf": op_gkr_cf, "gkr_wide": op_gkr_wide, "gkr_dense": op_gkr_dense, "gkr_sparse": op_gkr_sparse, "elementwise": op_elementwise,
       "fold": op_fold, "evaluate": op_evaluate, "basic_sumcheck": op_basic, "gkr_sumcheck": op_gkr_sumcheck, "msm": op_msm, "kzg": op_kzg}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--ops", default=",".join(o for o in OPS if not o.startswith("big_")))
    args = ap.parse_args()
    zk = G.import_package()
    from zkmle_amd import _lib
    _lib.check(zk.lib().zk_init(0))
    O.lib()
    rng = np.random.default_rng(args.seed)
    names = [o for o in args.ops.split(",") if o]
    counts = {o: 0 for o in names}
    t0 = time.time()
    last = t0
    while time.time() - t0 < args.seconds:
        name = names[int(rng.integers(0, len(names)))]
        field = FIELDS[int(rng.integers(0, len(FIELDS)))]
        mode = ["random", "random", "extreme", "max"][int(rng.integers(0, 4))]
        OPS[name](zk, rng, field, mode)
        counts[name] += 1
        if time.time() - last > 30:
            last = time.time()
            print(json.dumps({"elapsed_s": round(last - t0, 1), "cases": counts}), flush=True)
    print(json.dumps({"seed": args.seed, "seconds": args.seconds, "cases": counts, "panics": PANICS, "mismatches": 0}), flush=True)


if __name__ == "__main__":
    main()
