"""Child process of tests/test_gpu_variants.py: the provers under an environment switch (read once per process), checked against the
oracle.  Prints one JSON line {"checked": n, "mismatches": [...]}."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G                                                          # noqa: E402
from oracle import oracle as O                                                       # noqa: E402


def main():
    zk = G.import_package()
    from zkmle_amd import _lib
    _lib.check(zk.lib().zk_init(0))

    def rand_table(field, n, seed):
        t = np.zeros((n, zk.limbs(field)), np.uint64)
        assert zk.lib().zk_host_fill_random(field, seed, 0, n, t.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
        return t

    bad, checked = [], 0
    # basic sumcheck (prover.rs:35-71): sizes that take 0, 1 and several passes of 1..4 rounds before the tail
    for field, logns in ((0, (1, 3, 11, 12, 13, 14, 15, 16, 18, 19, 20, 21)), (1, (2, 12, 14, 15, 19)), (3, (13, 17, 20))):
        for logn in logns:
            table = rand_table(field, 1 << logn, 4000 + logn)
            prover = zk.Prover.init(field, table)
            proof = prover.prove()
            cs, rp, ch = O.sumcheck_basic_prove(field, table)
            ok = (np.array_equal(proof.initial_claimed_sum, cs) and np.array_equal(proof.round_univariate_polynomials.reshape(rp.shape), rp)
                  and np.array_equal(prover.challenges.reshape(ch.shape), ch) and zk.Verifier.init().verify(proof) is True)
            checked += 1
            if not ok:
                bad.append(["basic", field, logn])
    # the same prover through the sharded entry point on a one-rank communicator (zk_rounds_* handle)
    S = zk.sharded
    comm = S.Comm()
    for logn in (5, 12, 13, 16, 19, 20):
        table = rand_table(0, 1 << logn, 4100 + logn)
        cs, rp, ch = S.sumcheck_basic_prove_device(comm, S.GpuShard(zk.MultilinearPolynomial(0, table)), absorb_table=True)
        ecs, erp, ech = O.sumcheck_basic_prove(0, table)
        checked += 1
        if not (np.array_equal(cs, ecs) and np.array_equal(rp.reshape(erp.shape), erp) and np.array_equal(ch.reshape(ech.shape), ech)):
            bad.append(["sharded_basic", 0, logn])
    comm.close()
    # GKR sumcheck (sumcheck_gkr_protocol.rs:24-67)
    MP = zk.MultilinearPolynomial
    # (2^11 .. 2^19 with two products of two factors: every way into and out of the two-rounds-per-launch regime, sumcheck_kernels.cuh split2_round_kernel)
    for field, (nprod, nfac, logn) in ((0, (2, 2, 5)), (0, (2, 2, 11)), (0, (2, 2, 12)), (2, (3, 2, 13)), (0, (2, 3, 12)), (0, (2, 2, 16)), (2, (2, 2, 17)), (0, (2, 2, 19)), (0, (4, 2, 10))):
        n = 1 << logn
        tabs = np.stack([np.stack([rand_table(field, n, 60 * p + f + logn) for f in range(nfac)]) for p in range(nprod)])
        sp = zk.SumPolynomial([zk.ProductPolynomial([MP(field, t) for t in prod]) for prod in tabs])
        claimed = O.vec_sum(field, O.sumpoly_reduce(field, tabs))
        t_gpu, t_cpu = zk.Transcript(), O.Transcript()
        result = zk.sumcheck.prove(sp, claimed, t_gpu)
        co, chal = O.sumcheck_gkr_prove(field, tabs, claimed, t_cpu)
        checked += 1
        if not (np.array_equal(result.round_univariate_polynomials, co) and np.array_equal(result.random_challenges, chal)
                and t_gpu.sample_random_challenge() == t_cpu.sample_random_challenge()):
            bad.append(["gkr_sumcheck", field, nprod, nfac, logn])
    # sparse GKR proof (zkmle_gkr_sparse.hip) of a layered circuit wide enough for the half-table gate weights (> 12 output bits): its digest
    # must not depend on the code path (tests/test_gpu_variants.py compares the variants), and the sparse verifier must accept it
    import hashlib
    rng = np.random.default_rng(0xC1C)
    lg, depth = 15, 2                    # half a table = 2^14 entries: round 0 derives e(1) from the claim in the host-assisted mode only
    n = 1 << lg
    rows = []
    for _ in range(depth):
        g = np.zeros((n, 4), np.uint64)
        g[:, 0] = rng.integers(0, n, n); g[:, 1] = rng.integers(0, n, n); g[:, 2] = rng.integers(0, n, n); g[:, 3] = rng.integers(0, 2, n)
        rows.append(g)
    x = rand_table(0, n, 4242)
    proof = zk.gkr.sparse_prove(0, rows, [lg] * depth, x)
    h = hashlib.sha256()
    for arr in (proof.circuit_output, proof.claimed_sum, proof.layer_claims, proof.coeffs, proof.challenges, proof.wb_evals, proof.wc_evals,
                proof.output_challenges):
        h.update(np.ascontiguousarray(arr).tobytes())
    checked += 1
    if not zk.gkr.sparse_verify(0, rows, [lg] * depth, proof, x):
        bad.append(["sparse_gkr_verify", lg, depth])
    # the same with skewed wiring: thousands of gates under one left / one right index (several passes of the gate-parallel table kernels)
    lg2 = 13
    n2 = 1 << lg2
    rows2 = []
    for _ in range(2):
        g = np.zeros((n2, 4), np.uint64)
        g[:, 0] = rng.integers(0, n2, n2); g[:, 1] = rng.integers(0, n2, n2); g[:, 2] = rng.integers(0, n2, n2); g[:, 3] = rng.integers(0, 2, n2)
        g[: n2 // 2, 0] = 77
        g[n2 // 4: 3 * n2 // 4, 1] = 4321
        rows2.append(g)
    x2 = rand_table(0, n2, 4343)
    proof2 = zk.gkr.sparse_prove(0, rows2, [lg2] * 2, x2)
    for arr in (proof2.circuit_output, proof2.coeffs, proof2.challenges, proof2.wb_evals, proof2.wc_evals):
        h.update(np.ascontiguousarray(arr).tobytes())
    checked += 1
    if not zk.gkr.sparse_verify(0, rows2, [lg2] * 2, proof2, x2):
        bad.append(["sparse_gkr_verify_skewed", lg2])
    # the other fields through the same paths (the half tables in the products' internal form have a per-field limb count: 9 / 14 limbs of 29 bits)
    for fld in (1, 2, 3):
        x3 = rand_table(fld, n2, 4400 + fld)
        proof3 = zk.gkr.sparse_prove(fld, rows2, [lg2] * 2, x3)
        for arr in (proof3.circuit_output, proof3.coeffs, proof3.challenges, proof3.wb_evals, proof3.wc_evals):
            h.update(np.ascontiguousarray(arr).tobytes())
        checked += 1
        if not zk.gkr.sparse_verify(fld, rows2, [lg2] * 2, proof3, x3):
            bad.append(["sparse_gkr_verify_field", fld])
    # gkr_protocol::prove through the dense API (from the gate lists, or on the dense tables with ZK_GKR_DENSE_TABLES=1): the oracle's proof
    import random
    prng = random.Random(66)
    layers = []
    for i in range(6):
        n_in = 1 << (i + 1)
        layers.append([(prng.randrange(n_in), prng.randrange(n_in), o, prng.choice([0, 1])) for o in range(1 << i)])
    xin = rand_table(0, 1 << 6, 4500)
    circuit = zk.gkr.Circuit(0, [zk.gkr.Layer([zk.gkr.Gate(*g) for g in layer]) for layer in layers])
    gp = zk.gkr.prove(circuit, xin)
    want = O.gkr_prove(0, layers, xin)
    claims, co, ch = gp._flat
    checked += 1
    if not (np.array_equal(claims, want["layer_claims"]) and np.array_equal(co, want["coeffs"]) and np.array_equal(ch, want["challenges"])
            and np.array_equal(gp.wb_evaluations, want["wb_evals"]) and np.array_equal(gp.wc_evaluations, want["wc_evals"]) and zk.gkr.verify(circuit, gp, xin)):
        bad.append(["gkr_dense_api"])
    for arr in (claims, co, ch):
        h.update(np.ascontiguousarray(arr).tobytes())
    print(json.dumps({"checked": checked, "mismatches": bad, "sparse_gkr_digest": h.hexdigest(), "env": {k: v for k, v in os.environ.items() if k.startswith("ZK_")}}), flush=True)


if __name__ == "__main__":
    main()
