"""GPU: BASELINE config 5's 8-way split through the C-ABI provers (zk_sharded_*), bit-exact against the single-table oracle.

A one-GPU box allows at most 6 processes on its card, so the 8 ranks are 8 THREADS of this process, each with its own HIP
stream, exchanging through the library's ranks-as-threads transport (zk_comm_local_group_*, backend "local-threads"): the
same prover code, collectives staged through host memory instead of RCCL.  G = 8 reaches what the 2- and 4-rank tests do not:
three replicated rounds after the gather, the stride-8 interleave of the absorb and of the tail, an 8-point MSM combine and
the k = 3 leftover opening of zk_sharded_kzg_open.  (prover.rs:46-63, sumcheck_gkr_protocol.rs:37-60, evaluation_form.rs:21-33,
multilinear_kzg.rs:37-42,50-126.)"""
import ctypes as C

import numpy as np
import pytest

import __graft_entry__ as G
from oracle import oracle as O

from test_local_group_cpu import run_ranks

pytestmark = pytest.mark.gpu
WORLD = 8


def fill(zk, field, n, seed):
    from zkmle_amd import _lib
    t = np.zeros((n, zk.limbs(field)), np.uint64)
    _lib.check(zk.lib().zk_host_fill_random(field, seed, 0, n, _lib.p64(t)))
    return t


def on_own_stream(zk, body):
    """rank thread prologue: this thread's device and a stream of its own (every launch of the library follows it)"""
    import torch
    from zkmle_amd import _lib

    def wrapped(rank, comm):
        lib = zk.lib()
        _lib.check(lib.zk_init(0))
        st = torch.cuda.Stream()
        lib.zk_set_stream.argtypes = [C.c_void_p]
        _lib.check(lib.zk_set_stream(C.c_void_p(st.cuda_stream)))
        try:
            return body(rank, comm)
        finally:
            lib.zk_set_stream(None)
    return wrapped


@pytest.mark.parametrize("field", [O.FR381, O.BN254_FR])
def test_eight_way_sumcheck_provers_vs_oracle(field):
    """Prover::prove with the whole-table absorb on ONE 2^20 table, the GKR sumcheck on 4 tables of 2^16 and evaluate, 8 ranks"""
    zk = G.import_package()
    S = zk.sharded
    logn = 20 if field == O.FR381 else 14
    n, ns = 1 << logn, 1 << 16
    table = fill(zk, field, n, 0xC5_0001)
    sum_tables = np.stack([np.stack([fill(zk, field, ns, 0xC5_0100 + 2 * p + f) for f in range(2)]) for p in range(2)])
    claimed = O.vec_sum(field, O.sumpoly_reduce(field, sum_tables))
    point = fill(zk, field, logn, 0xC5_0200)

    def body(rank, comm):
        res = {"backend": comm.native_backend()}
        shard = S.GpuShard.from_array(field, S.shard_of(table, rank, WORLD))
        rx0, _ = comm.native_stats()
        cs0, rp0, ch0 = S.sumcheck_basic_prove_device(comm, shard, absorb_table=False)
        rx1, nc1 = comm.native_stats()
        cs, rp, ch = S.sumcheck_basic_prove_device(comm, shard)
        rx2, nc2 = comm.native_stats()
        res.update(claimed=cs, rounds=rp, chal=ch, noabsorb_claimed=cs0, noabsorb_rounds0=rp0[0],
                   absorb_rx=(rx2 - rx1) - (rx1 - rx0), colls_noabsorb=nc1)
        t = zk.Transcript()
        t.append(b"prefix")
        ss = S.GpuSumShard(field, [[zk.MultilinearPolynomial(field, np.ascontiguousarray(tt[rank::WORLD])) for tt in prod] for prod in sum_tables])
        co, gch, fin = S.sumcheck_gkr_prove_device(comm, ss, claimed, t)
        res.update(gkr_coeffs=co, gkr_chal=gch, gkr_final=fin, gkr_tail=np.frombuffer(t.sample_random_challenge(), np.uint8))
        res["evaluate"] = S.mle_evaluate(comm, shard.poly, point)
        return res

    outs = run_ranks(WORLD, on_own_stream(zk, body))
    ecs, erp, ech = O.sumcheck_basic_prove(field, table)
    et = O.Transcript()
    et.append(b"prefix")
    eco, egch = O.sumcheck_gkr_prove(field, sum_tables, claimed, et)
    etail = np.frombuffer(et.sample_random_challenge(), np.uint8)
    efin = np.stack([O.evaluate(field, sum_tables[p, f], egch) for p in range(2) for f in range(2)])
    eev = O.evaluate(field, table, point)
    for rank, res in enumerate(outs):
        assert res["backend"] == "local-threads"
        assert np.array_equal(res["claimed"], ecs) and np.array_equal(res["rounds"], erp) and np.array_equal(res["chal"], ech), rank
        assert np.array_equal(res["noabsorb_claimed"], ecs) and np.array_equal(res["noabsorb_rounds0"], erp[0])
        assert np.array_equal(res["gkr_coeffs"], eco) and np.array_equal(res["gkr_chal"], egch), rank
        assert np.array_equal(res["gkr_final"], efin) and np.array_equal(res["gkr_tail"], etail), rank
        assert np.array_equal(res["evaluate"], eev), rank
        if rank:                                      # the absorb costs the non-root ranks the 208-byte sponge, nothing table-sized
            assert res["absorb_rx"] == 208, (rank, res["absorb_rx"])
    # 2^20 / 8 = 2^17 per rank: (20 - 11) rounds in passes of <= 4 -> 3 all-reduces + 1 all-gather of the last 2048 entries
    assert outs[0]["colls_noabsorb"] <= 5 or field != O.FR381


def test_eight_way_msm_and_kzg_open_vs_oracle():
    """commit_to_polynomial and open_and_prove of a 2^10 table on a real trusted setup, table and G1 powers low-bit-sharded 8 ways"""
    zk = G.import_package()
    S = zk.sharded
    logn = 10
    n = 1 << logn
    table = fill(zk, 0, n, 0xC5_0300)
    taus = fill(zk, 0, logn, 0xC5_0301)
    opening = fill(zk, 0, logn, 0xC5_0302)
    pts = O.kzg_setup_g1(taus)
    want_ev, want_proofs = O.kzg_open(table, pts, opening)
    want_commit = O.kzg_commit(table, pts)

    def body(rank, comm):
        poly = zk.MultilinearPolynomial.vector(0, np.ascontiguousarray(table[rank::WORLD]))
        bases = zk.G1Bases(np.ascontiguousarray(pts[rank::WORLD]))
        ev, proofs = S.kzg_open_device(comm, poly, bases, opening)
        commit = S.msm_device(comm, poly, bases)
        lo, hi = rank * n // WORLD, (rank + 1) * n // WORLD           # the slice partition gives the same point
        commit2 = S.msm_device(comm, zk.MultilinearPolynomial.vector(0, table[lo:hi]), zk.G1Bases(pts[lo:hi]))
        return ev, proofs, commit, commit2

    for rank, (ev, proofs, commit, commit2) in enumerate(run_ranks(WORLD, on_own_stream(zk, body))):
        assert np.array_equal(ev, want_ev), rank
        assert np.array_equal(proofs, want_proofs), rank
        assert np.array_equal(commit, want_commit) and np.array_equal(commit2, want_commit), rank


def test_one_rank_whose_host_side_stalls_does_not_hang_the_others():
    """Fault injection through the sharded prover (2 ranks as threads): the host side of ONE rank's proof is deaf for 3.5 s
    (zk_debug_stall_service_once).  Its kernels give up and end, the rank keeps its place in every collective, BOTH ranks return within
    seconds with an error -- the stalled one naming the cause, the other one told by the status word the ranks sum at the end -- and the
    next sharded proof on the same threads equals the oracle's."""
    import os
    import time
    zk = G.import_package()
    from zkmle_amd import _lib
    os.environ["ZK_ENABLE_FAULT_INJECTION"] = "1"             # the test hook is process-global: opt in (include/zkmle.h)
    S = zk.sharded
    field, world, logn = O.FR381, 2, 16
    table = fill(zk, field, 1 << logn, 0xC5_0400)
    want = O.sumcheck_basic_prove(field, table)

    def body(rank, comm):
        shard = S.GpuShard.from_array(field, S.shard_of(table, rank, world))
        S.sumcheck_basic_prove_device(comm, shard)                 # warm-up: every rank has its service thread
        comm.barrier()
        if rank == 0:
            _lib.check(zk.lib().zk_debug_stall_service_once(3500))
        comm.barrier()
        t0 = time.time()
        err = None
        try:
            S.sumcheck_basic_prove_device(comm, shard)
        except zk.ZkError as e:
            err = str(e)
        took = time.time() - t0
        comm.barrier()
        cs, rp, ch = S.sumcheck_basic_prove_device(comm, shard)
        return err, took, cs, rp, ch

    try:
        outs = run_ranks(world, on_own_stream(zk, body))
    finally:
        os.environ.pop("ZK_ENABLE_FAULT_INJECTION", None)
    errs = [o[0] for o in outs]
    assert all(e is not None for e in errs), ("the ranks must agree that the proof failed", errs)   # one status word summed at the end
    assert any("host" in e.lower() for e in errs) and all("host" in e.lower() or "another rank" in e.lower() for e in errs), errs
    for err, took, cs, rp, ch in outs:
        assert took < 15.0, took
        assert np.array_equal(cs, want[0]) and np.array_equal(rp, want[1]) and np.array_equal(ch, want[2])
