"""GPU: the multi-GPU provers with two ranks sharing cuda:0 (HIP kernels per shard, gloo exchange), and
prove_succinct, bit-exact against the oracle."""
import numpy as np
import pytest

import __graft_entry__ as G
from oracle import oracle as O

from test_sharded_cpu import launch, expected, check, rand_table

pytestmark = pytest.mark.gpu
OPS = {"add": 0, "mul": 1}


@pytest.mark.parametrize("world,logn", [(1, 3), (2, 1), (2, 2), (2, 10), (4, 2), (4, 9), (2, 14), (2, 17)])
def test_sharded_provers_on_gpu(world, logn):
    field = O.FR381
    n = 1 << logn
    table = rand_table(field, n, 20 + logn)
    sum_tables = np.stack([np.stack([rand_table(field, n, 100 * p + 10 * f + logn) for f in range(2)]) for p in range(2)])
    claimed = O.vec_sum(field, O.sumpoly_reduce(field, sum_tables))
    g = O.g1_generator()
    pts = np.stack([O.g1_mul_fr(g, O.from_ints(O.FR381, [7 + 3 * i])[0]) for i in range(16)])
    scalars = rand_table(O.FR381, 16, 78)
    results = launch(world, "gpu", field, table, sum_tables, claimed, scalars, pts)
    want = expected(field, table, sum_tables, claimed)
    want["msm"] = O.kzg_commit(scalars, pts)
    # device-resident rounds (zk_rounds: limb all-reduce + transcript on the GPU) give the same proofs
    for k in ("basic_claimed", "basic_rounds", "basic_chal", "gkr_coeffs", "gkr_chal", "gkr_tail"):
        want["dev_" + k] = want[k]
    want["dev_gkr_final"] = np.stack([O.evaluate(field, sum_tables[p, f], want["gkr_chal"]) for p in range(2) for f in range(2)])
    want["dev_noabsorb_claimed"] = want["basic_claimed"]
    want["dev_msm"] = want["msm"]
    want["dev_evaluate"] = O.evaluate(field, table, table[:logn])
    want["dev_backend"] = np.frombuffer(b"host-ops", np.uint8)
    check(results, want)


def test_prove_succinct_reference_circuits(ref_kats):
    zk = G.import_package()
    from zkmle_amd import _lib
    _lib.check(zk.lib().zk_init(0))
    for k in ref_kats["succinct_gkr_roundtrip"]:              # succinct_gkr_protocol.rs:295-360
        spec = k["layers"]
        circuit = zk.Circuit.new(0, [zk.Layer.new([zk.Gate.new(g[0], g[1], g[2], OPS[g[3]]) for g in layer]) for layer in spec])
        inputs = zk.from_ints(0, k["inputs"])
        taus = zk.from_ints(0, k["taus"])
        setup = zk.TrustedSetup.initialize_setup(taus)
        proof = zk.gkr.prove_succinct(circuit, inputs, setup)
        pts = O.kzg_setup_g1(taus)
        layers = [[(g[0], g[1], g[2], OPS[g[3]]) for g in layer] for layer in spec]
        want = O.gkr_prove_succinct(layers, inputs, pts)
        assert np.array_equal(proof.claimed_sum, want["claimed_sum"])
        assert np.array_equal(proof._flat[1], want["coeffs"])
        assert np.array_equal(proof.input_polynomial_commitment, want["input_polynomial_commitment"])
        for got, (ev, prs) in ((proof.input_rb_proof, want["input_rb_proof"]), (proof.input_rc_proof, want["input_rc_proof"])):
            assert np.array_equal(got.evaluation, ev) and np.array_equal(got.proofs, prs)
        # verify_succinct (succinct_gkr_protocol.rs:172-285, the reference's own round trip :295-360): GKR rounds + 2 KZG verifications
        assert zk.gkr.verify_succinct(circuit, proof, setup) is True
        bad = zk.MultilinearKZGProof(proof.input_rb_proof.evaluation, proof.input_rb_proof.proofs[::-1].copy())
        tampered = zk.gkr.SuccinctProof(proof, proof.input_polynomial_commitment, bad, proof.input_rc_proof)
        if len(bad.proofs) > 1 and not np.array_equal(bad.proofs, proof.input_rb_proof.proofs):
            assert zk.gkr.verify_succinct(circuit, tampered, setup) is False
        # the opened values are the verifier's wb / wc of the input layer (succinct_gkr_protocol.rs:226-233)
        ch = want["challenges"][-O.gkr_rounds(len(layers) - 1):]
        assert np.array_equal(proof.input_rb_proof.evaluation, O.evaluate(O.FR381, inputs, ch[: len(ch) // 2]))


@pytest.mark.parametrize("logn", [3, 13])
def test_rccl_world_size_1_native_provers(logn):
    """process group "nccl" (= RCCL) with one rank on the one GPU: the library's own RCCL communicator runs every collective
    of the sharded provers on the prover's stream; proofs equal the oracle's.  (RCCL refuses two ranks on one device, so
    more ranks than GPUs go over the exchange callbacks: test_sharded_provers_on_gpu.)"""
    import os
    import tempfile
    import torch.multiprocessing as mp
    from test_sharded_cpu import free_port
    from _sharded_workers import run_rccl_world1
    field = O.FR381
    n = 1 << logn
    table = rand_table(field, n, 40 + logn)
    sum_tables = np.stack([np.stack([rand_table(field, n, 300 * p + 10 * f + logn) for f in range(2)]) for p in range(2)])
    claimed = O.vec_sum(field, O.sumpoly_reduce(field, sum_tables))
    g = O.g1_generator()
    pts = np.stack([O.g1_mul_fr(g, O.from_ints(O.FR381, [11 + 3 * i])[0]) for i in range(8)])
    scalars = rand_table(O.FR381, 8, 79)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(run_rccl_world1, args=(free_port(), field, table, sum_tables, claimed, scalars, pts, d), nprocs=1, join=True)
        res = dict(np.load(os.path.join(d, "rank0.npz")))
    assert res["backend"].tobytes() == b"rccl"
    assert np.array_equal(res["limbs"], np.arange(27, dtype=np.int64) * 0x1_0000_0001)       # sum over one rank
    assert np.array_equal(res["points"][0], pts[:2].reshape(-1))
    want = expected(field, table, sum_tables, claimed)
    for k in ("basic_claimed", "basic_rounds", "basic_chal", "gkr_coeffs", "gkr_chal", "gkr_tail"):
        assert np.array_equal(res[k], want[k]), k
    assert np.array_equal(res["gkr_final"], np.stack([O.evaluate(field, sum_tables[p, f], want["gkr_chal"]) for p in range(2) for f in range(2)]))
    assert np.array_equal(res["msm"], O.kzg_commit(scalars, pts))
    assert np.array_equal(res["evaluate"], O.evaluate(field, table, table[:logn]))
    assert int(res["ncoll"][0]) >= 3


@pytest.mark.parametrize("world,logn,backend", [(1, 3, "gloo"), (2, 1, "gloo"), (2, 4, "gloo"), (4, 2, "gloo"), (4, 5, "gloo"), (2, 6, "gloo"), (1, 4, "nccl")])
def test_sharded_kzg_open_and_commit(world, logn, backend):
    """open_and_prove / commit_to_polynomial (multilinear_kzg.rs:25-126) with the table and the setup's G1 powers low-bit-sharded over
    the ranks (zk_sharded_kzg_open, zk_sharded_msm_g1): evaluation, every proof point and the commitment equal the oracle's naive
    single-device opening; the "nccl" case runs the exchange over the library's own RCCL communicator (one rank)."""
    import os
    import tempfile
    import torch.multiprocessing as mp
    from test_sharded_cpu import free_port
    from _sharded_workers import run_kzg_open
    n = 1 << logn
    table = rand_table(O.FR381, n, 500 + logn)
    taus = rand_table(O.FR381, logn, 600 + logn)
    pts = O.kzg_setup_g1(taus)
    opening = rand_table(O.FR381, logn, 700 + logn)
    want_ev, want_proofs = O.kzg_open(table, pts, opening)
    want_commit = O.kzg_commit(table, pts)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(run_kzg_open, args=(world, free_port(), backend, table, pts, opening, d), nprocs=world, join=True)
        results = [dict(np.load(os.path.join(d, f"rank{r}.npz"))) for r in range(world)]
    for res in results:
        assert np.array_equal(res["ev"], want_ev)
        assert np.array_equal(res["proofs"], want_proofs)
        assert np.array_equal(res["commit"], want_commit)
        assert res["backend"].tobytes() == (b"rccl" if backend == "nccl" else b"host-ops")
