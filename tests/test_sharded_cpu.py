"""world_size-2 gloo test (CPU) of the multi-GPU host logic: low-bit sharding, per-round all-gather of
partial sums, replicated tail, slice-sharded MSM.  The per-shard compute is an oracle-backed test double
here (no GPU in this container); tests/test_gpu_sharded.py runs the same workers on HIP kernels."""
import os
import socket
import tempfile

import numpy as np
import pytest

from oracle import oracle as O

from _sharded_workers import run


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rand_table(field, n, seed):
    rng = np.random.default_rng(seed)
    p = O.modulus(field)
    return O.from_ints(field, [int.from_bytes(rng.bytes(40), "little") % p for _ in range(n)])


def launch(world, engine, field, table, sum_tables, claimed, scalars=None, points=None):
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(run, args=(world, free_port(), engine, field, table, sum_tables, claimed, scalars, points, d), nprocs=world, join=True)
        return [dict(np.load(os.path.join(d, f"rank{r}.npz"))) for r in range(world)]


def expected(field, table, sum_tables, claimed):
    cs, rp, ch = O.sumcheck_basic_prove(field, table)
    t = O.Transcript()
    t.append(b"prefix")
    co, gch = O.sumcheck_gkr_prove(field, sum_tables, claimed, t)
    ta = O.Transcript()
    ta.append(O.mle_to_bytes(field, table))
    return dict(basic_claimed=cs, basic_rounds=rp, basic_chal=ch, gkr_coeffs=co, gkr_chal=gch,
                gkr_tail=np.frombuffer(t.sample_random_challenge(), np.uint8),
                absorb_digest=np.frombuffer(ta.sample_random_challenge(), np.uint8))


def check(results, want):
    for res in results:                       # every rank holds the same, reference-identical proof
        for k, v in want.items():
            assert np.array_equal(res[k], v), k
    # the whole-table absorb is not a bandwidth-sized collective for anyone but the hashing rank: the others receive the
    # 208-byte sponge (25 lanes + fill) and nothing else
    for key in ("absorb_rx", "dev_absorb_rx"):
        for rank, res in enumerate(results):
            if key in res and len(results) > 1 and rank != 0:
                assert int(res[key][0]) == 208, (key, rank, res[key])


@pytest.mark.parametrize("logn", [1, 2, 3, 6])
def test_sharded_provers_two_ranks_gloo(logn):
    field = O.FR381
    n = 1 << logn
    table = rand_table(field, n, 10 + logn)
    sum_tables = np.stack([np.stack([rand_table(field, n, 100 * p + 10 * f + logn) for f in range(2)]) for p in range(2)])
    claimed = O.vec_sum(field, O.sumpoly_reduce(field, sum_tables))
    g = O.g1_generator()
    pts = np.stack([O.g1_mul_fr(g, O.from_ints(O.FR381, [3 + 5 * i])[0]) for i in range(6)])
    scalars = rand_table(O.FR381, 6, 77)
    results = launch(2, "oracle", field, table, sum_tables, claimed, scalars, pts)
    want = expected(field, table, sum_tables, claimed)
    want["msm"] = O.kzg_commit(scalars, pts)
    check(results, want)


def test_shard_layout_is_low_bit():
    import __graft_entry__ as G
    S = G.import_package().sharded
    t = np.arange(16 * 4, dtype=np.uint64).reshape(16, 4)
    for world in (1, 2, 4, 8):
        shards = [S.shard_of(t, r, world) for r in range(world)]
        back = np.stack(shards, axis=1).reshape(16, 4)      # element j of rank r is global j * world + r
        assert np.array_equal(back, t)
