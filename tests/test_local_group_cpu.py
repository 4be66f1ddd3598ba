"""CPU: the ranks-as-threads transport (include/zkmle.h zk_comm_local_group_*, zk_comm_host_exchange) and BASELINE config 5's
8-way split of the host-driven protocol model (tests/_sharded_protocol_model.py) over it.  The per-shard compute is the oracle-backed test double of tests/_sharded_workers.py
(no GPU here); tests/test_gpu_config5_8way.py runs the same split through the C-ABI provers on HIP kernels."""
import threading

import numpy as np
import pytest

import __graft_entry__ as G
from oracle import oracle as O

import _sharded_protocol_model as M
from _sharded_workers import OracleShard, OracleSumShard
from test_sharded_cpu import expected, rand_table


def run_ranks(world, body):
    """body(rank, comm) on `world` threads; returns the per-rank results, re-raises the first failure"""
    S = G.import_package().sharded
    group = S.LocalGroup(world)
    out, errs = [None] * world, []

    def main(rank):
        comm = None
        try:
            comm = group.comm(rank)
            out[rank] = body(rank, comm)
        except BaseException as e:                      # noqa: BLE001
            errs.append((rank, e))
            group.abort()
        finally:
            if comm is not None:
                comm.close()

    ts = [threading.Thread(target=main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not any(t.is_alive() for t in ts), "a rank hangs in an exchange"
    group.close()
    if errs:
        raise errs[0][1]
    return out


@pytest.mark.parametrize("world", [1, 2, 8])
def test_local_group_exchanges(world):
    def body(rank, comm):
        assert comm.native_backend() == "local-threads"
        res = []
        for it in range(200):                           # back-to-back exchanges reuse the published slots: no stale reads
            a = comm.all_reduce_sum_i64(np.arange(27, dtype=np.int64) * (rank + 1) + it)
            g = comm.all_gather(np.full(5, 1000 * it + rank, np.uint64))
            parts = comm.gather_bytes(bytes([rank, it % 251]) * 3, root=it % world)
            b = comm.broadcast_u64(np.full(26, 7 * it + rank, np.uint64), root=(it + 1) % world)
            res.append((a, g, parts, b))
        return res

    outs = run_ranks(world, body)
    tri = world * (world + 1) // 2
    for rank, res in enumerate(outs):
        for it, (a, g, parts, b) in enumerate(res):
            assert np.array_equal(a, np.arange(27, dtype=np.int64) * tri + it * world)
            assert np.array_equal(g, np.stack([np.full(5, 1000 * it + r, np.uint64) for r in range(world)]))
            if rank == it % world:
                assert parts == [bytes([r, it % 251]) * 3 for r in range(world)]
            else:
                assert parts is None
            assert np.array_equal(b, np.full(26, 7 * it + (it + 1) % world, np.uint64))


def test_local_group_abort_wakes_waiting_ranks():
    """a rank that fails calls abort(): the ranks waiting for it in an exchange return ZK_E_COMM instead of hanging"""
    def body(rank, comm):
        if rank == 3:
            raise RuntimeError("rank 3 failed before the exchange")
        comm.all_gather(np.zeros(4, np.uint64))          # would wait for rank 3 forever
        return "unreachable"

    with pytest.raises(RuntimeError, match="rank 3 failed"):
        run_ranks(4, body)


@pytest.mark.parametrize("logn", [3, 5, 8])
def test_config5_split_eight_ranks_host_flow(logn):
    """the 8-way low-bit split (3 replicated rounds after the gather, stride-8 interleave of the absorb) == the single-table oracle"""
    S = G.import_package().sharded
    zk = G.import_package()
    field, world = O.FR381, 8
    n = 1 << logn
    table = rand_table(field, n, 900 + logn)
    sum_tables = np.stack([np.stack([rand_table(field, n, 950 + 100 * p + 10 * f + logn) for f in range(2)]) for p in range(2)])
    claimed = O.vec_sum(field, O.sumpoly_reduce(field, sum_tables))

    def body(rank, comm):
        res = {}
        cs, rp, ch = M.sumcheck_basic_prove(comm, OracleShard(field, S.shard_of(table, rank, world)))
        res.update(basic_claimed=cs, basic_rounds=rp, basic_chal=ch)
        t = zk.Transcript()
        t.append(b"prefix")
        co, gch = M.sumcheck_gkr_prove(comm, OracleSumShard(field, sum_tables[:, :, rank::world]), claimed, t)
        res.update(gkr_coeffs=co, gkr_chal=gch, gkr_tail=np.frombuffer(t.sample_random_challenge(), np.uint8))
        return res

    want = expected(field, table, sum_tables, claimed)
    want.pop("absorb_digest")
    for res in run_ranks(world, body):
        for k, v in want.items():
            assert np.array_equal(res[k], v), k
