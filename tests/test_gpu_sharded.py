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


def test_rounds_handle_several_rounds_per_pass_and_its_argument_checks():
    """zk_rounds_multi_* (include/zkmle.h): a host driving its own loop gets the oracle's proof, and misuse is refused with ZK_E_ARG"""
    import ctypes as C
    import numpy as np
    import __graft_entry__ as G
    from oracle import oracle as O
    zk = G.import_package()
    from zkmle_amd import _lib as L
    lib = zk.lib()
    L.check(lib.zk_init(0))
    vp, u64p = C.c_void_p, L.u64p
    lib.zk_rounds_new.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, vp, C.POINTER(vp)]
    lib.zk_rounds_free.argtypes = [vp]
    lib.zk_rounds_limbs_len.argtypes = [vp]
    lib.zk_rounds_limbs_len.restype = C.c_size_t
    lib.zk_rounds_multi_max.argtypes = [vp]
    lib.zk_rounds_multi_max.restype = C.c_uint
    lib.zk_rounds_multi_evals.argtypes = [vp, vp, C.c_uint, vp]
    lib.zk_rounds_multi_absorb.argtypes = [vp, vp, C.c_uint]
    lib.zk_rounds_multi_fold_evals.argtypes = [vp, vp, vp, C.c_uint, C.c_uint, vp]
    lib.zk_rounds_multi_tail.argtypes = [vp, vp]
    lib.zk_rounds_collect.argtypes = [vp, vp, u64p, u64p, u64p, u64p]
    field, logn = 0, 15
    n = 1 << logn
    poly = zk.MultilinearPolynomial.random(field, n, 0xABC)
    table = poly.evaluated_values
    tr = zk.Transcript()
    tr.append(O.mle_to_bytes(field, table))                       # prover.rs:38-39, as the caller of the handle does it
    r = vp()
    L.check(lib.zk_rounds_new(field, 0, 1, 1, logn, tr._h, C.byref(r)))
    kmax = lib.zk_rounds_multi_max(r)
    if kmax == 0:                                                 # ZK_HOST_TRANSCRIPT=0 in the environment: the one-round sequence applies
        lib.zk_rounds_free(r)
        pytest.skip("the transcript step runs on the device in this environment")
    assert kmax == 7                                              # ZK_BASIC_ROUNDS_PER_PASS default (kMultiMax = 8 is the ceiling)
    assert lib.zk_rounds_limbs_len(r) * 8 <= 1024 * 32
    limbs = zk.MultilinearPolynomial.alloc(field, 1024)           # device scratch for the limb words
    lp = lib.zk_table_device_ptr(limbs._h)
    t1 = zk.MultilinearPolynomial.alloc(field, n >> 3)
    t2 = zk.MultilinearPolynomial.alloc(field, n >> 4)
    E = L.ZK_E_ARG
    assert lib.zk_rounds_multi_evals(r, poly._h, 9, lp) == E          # more rounds than the handle runs per pass
    assert lib.zk_rounds_multi_evals(r, poly._h, 0, lp) == E
    assert lib.zk_rounds_multi_fold_evals(r, poly._h, t1._h, 3, 0, lp) == E   # nothing absorbed yet
    assert lib.zk_rounds_multi_tail(r, poly._h) == E                  # too long for the tail
    L.check(lib.zk_rounds_multi_evals(r, poly._h, 3, lp))             # 15 rounds = 3 + 1 + 11
    L.check(lib.zk_rounds_multi_absorb(r, lp, 3))
    assert lib.zk_rounds_multi_fold_evals(r, poly._h, t1._h, 4, 0, lp) == E   # only 3 rounds absorbed
    assert lib.zk_rounds_multi_fold_evals(r, poly._h, poly._h, 3, 0, lp) == E # in place
    L.check(lib.zk_rounds_multi_fold_evals(r, poly._h, t1._h, 3, 1, lp))
    L.check(lib.zk_rounds_multi_absorb(r, lp, 1))
    L.check(lib.zk_rounds_multi_fold_evals(r, t1._h, t2._h, 1, 0, None))
    assert lib.zk_rounds_multi_tail(r, t1._h) == E                    # wrong table (too long, and the round count would not add up)
    L.check(lib.zk_rounds_multi_tail(r, t2._h))
    nl = zk.limbs(field)
    cs, rp, ch = np.zeros(nl, np.uint64), np.zeros((logn, 2, nl), np.uint64), np.zeros((logn, nl), np.uint64)
    L.check(lib.zk_rounds_collect(r, tr._h, L.p64(cs), L.p64(rp), L.p64(ch), None))
    lib.zk_rounds_free(r)
    ecs, erp, ech = O.sumcheck_basic_prove(field, table)
    assert np.array_equal(cs, ecs) and np.array_equal(rp.reshape(erp.shape), erp) and np.array_equal(ch.reshape(ech.shape), ech)
    assert np.array_equal(poly.evaluated_values, table)               # the caller's table is never folded in place
    # a GKR-sumcheck handle (mode 1) does not offer it
    r2 = vp()
    t2r = zk.Transcript()
    L.check(lib.zk_rounds_new(field, 1, 2, 2, 4, t2r._h, C.byref(r2)))
    assert lib.zk_rounds_multi_max(r2) == 0
    assert lib.zk_rounds_multi_evals(r2, poly._h, 2, lp) == E
    lib.zk_rounds_free(r2)
