"""Worker bodies for the world_size-2 tests of the multi-GPU host logic (spawned by torch.multiprocessing).

`engine="oracle"` runs the per-shard compute on the CPU oracle (a TEST DOUBLE for the HIP engine, so the
sharding / exchange / transcript logic is covered where there is no GPU); `engine="gpu"` uses the HIP-backed
adapters of tests/_sharded_protocol_model.py for the host-driven flow AND the product's one-call C-ABI provers, both ranks on cuda:0.  The process group is gloo in both cases; on a real
multi-GPU node bench.py uses "nccl" (= RCCL)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class OracleShard:
    """test double of _sharded_protocol_model.HipShard on the CPU oracle"""

    def __init__(self, field, arr):
        self.field, self.arr = field, np.ascontiguousarray(arr, np.uint64)

    def spawn(self, arr):
        return OracleShard(self.field, arr)

    def __len__(self):
        return self.arr.shape[0]

    def half_sums(self):
        from oracle import oracle as O
        return O.split_and_sum(self.field, self.arr)

    def fold(self, r):
        from oracle import oracle as O
        return OracleShard(self.field, O.partial_evaluate(self.field, self.arr, 0, r))

    def fold_half_sums(self, r):
        f = self.fold(r)
        return f, f.half_sums()

    def download(self):
        return self.arr

    def to_bytes(self):
        from oracle import oracle as O
        return O.mle_to_bytes(self.field, self.arr)


class OracleSumShard:
    def __init__(self, field, tabs):
        self.field, self.tabs = field, np.ascontiguousarray(tabs, np.uint64)   # (nprod, nfac, len, limbs)
        self.nprod, self.nfac = self.tabs.shape[0], self.tabs.shape[1]

    def spawn(self, arrays):
        return OracleSumShard(self.field, arrays)

    def __len__(self):
        return self.tabs.shape[2]

    def round_evals(self):
        from oracle import oracle as O
        return O.gkr_round_univariate(self.field, self.tabs)

    def fold(self, r):
        from oracle import oracle as O
        return OracleSumShard(self.field, np.stack([np.stack([O.partial_evaluate(self.field, t, 0, r) for t in prod]) for prod in self.tabs]))

    def fold_round_evals(self, r):
        f = self.fold(r)
        return f, f.round_evals()

    def download(self):
        return self.tabs


def run(rank, world, port, engine, field, table, sum_tables, claimed, scalars, points, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as G
        zk = G.import_package()
        S = zk.sharded
        import _sharded_protocol_model as M
        comm = S.Comm()
        if engine == "gpu":
            from zkmle_amd import _lib
            _lib.check(zk.lib().zk_init(0))
            mk = lambda a: M.HipShard.from_array(field, a)
            mk_sum = lambda a: M.HipSumShard(field, [[zk.MultilinearPolynomial(field, t) for t in prod] for prod in a])
        else:
            mk = lambda a: OracleShard(field, a)
            mk_sum = lambda a: OracleSumShard(field, a)
        res = {}
        cs, rp, ch = M.sumcheck_basic_prove(comm, mk(S.shard_of(table, rank, world)))
        res.update(basic_claimed=cs, basic_rounds=rp, basic_chal=ch)
        t = zk.Transcript()
        t.append(b"prefix")
        co, gch = M.sumcheck_gkr_prove(comm, mk_sum(sum_tables[:, :, rank::world]), claimed, t)
        res.update(gkr_coeffs=co, gkr_chal=gch, gkr_tail=np.frombuffer(t.sample_random_challenge(), np.uint8))
        # the whole-table absorb alone (prover.rs:38-39): rank 0 hashes a streamed gather, the others get the 208-byte sponge
        tb = mk(S.shard_of(table, rank, world)).to_bytes()
        t = zk.Transcript()
        rx0 = comm.bytes_received
        M.absorb_sharded_table(comm, t, tb, 8 * zk.limbs(field), chunk_elems=3)
        res.update(absorb_rx=np.array([comm.bytes_received - rx0]), absorb_digest=np.frombuffer(t.sample_random_challenge(), np.uint8))
        if engine == "gpu":        # the product path: the same proofs through the C-ABI provers (zk_sharded_*), gloo as exchange callbacks
            shard = mk(S.shard_of(table, rank, world))
            rx0, _ = comm.native_stats()
            cs, rp, ch = S.sumcheck_basic_prove_device(comm, shard, absorb_table=False)
            rx1, _ = comm.native_stats()
            cs2, rp2, ch2 = S.sumcheck_basic_prove_device(comm, shard)
            rx2, ncoll = comm.native_stats()
            res.update(dev_basic_claimed=cs2, dev_basic_rounds=rp2, dev_basic_chal=ch2, dev_noabsorb_claimed=cs,
                       dev_absorb_rx=np.array([(rx2 - rx1) - (rx1 - rx0)]), dev_backend=np.frombuffer(comm.native_backend().encode(), np.uint8))
            t = zk.Transcript()
            t.append(b"prefix")
            co, gch, fin = S.sumcheck_gkr_prove_device(comm, mk_sum(sum_tables[:, :, rank::world]), claimed, t)
            res.update(dev_gkr_coeffs=co, dev_gkr_chal=gch, dev_gkr_final=fin,
                       dev_gkr_tail=np.frombuffer(t.sample_random_challenge(), np.uint8))
            nv = (table.shape[0]).bit_length() - 1
            if nv >= 1:
                res["dev_evaluate"] = S.mle_evaluate(comm, shard.poly, table[:nv])
        if scalars is not None:
            n = scalars.shape[0]
            lo, hi = rank * n // world, (rank + 1) * n // world           # slice sharding of the MSM terms
            if engine == "gpu":
                local = lambda: zk.kzg.msm(zk.MultilinearPolynomial.vector(0, scalars[lo:hi]), zk.G1Bases(points[lo:hi]))
            else:
                from oracle import oracle as O
                local = lambda: O.kzg_commit(scalars[lo:hi], points[lo:hi])
            res["msm"] = M.msm(comm, local)
            if engine == "gpu":
                res["dev_msm"] = S.msm_device(comm, zk.MultilinearPolynomial.vector(0, scalars[lo:hi]), zk.G1Bases(points[lo:hi]))
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    finally:
        dist.destroy_process_group()


def run_rccl_world1(rank, port, field, table, sum_tables, claimed, scalars, points, out_dir):
    """The RCCL leg on the hardware a one-GPU box has: process group "nccl" with world_size 1 on cuda:0.  Every collective of
    the C-ABI provers goes through ncclAllReduce / ncclAllGather / ncclBroadcast of a communicator the library created
    itself (zk_comm_init_rccl), on the prover's stream; the torch-level nccl group carries the Python-level exchanges."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import __graft_entry__ as G
        zk = G.import_package()
        from zkmle_amd import _lib
        _lib.check(zk.lib().zk_init(0))
        S = zk.sharded
        comm = S.Comm(device=torch.device("cuda", 0))
        res = {"backend": np.frombuffer(comm.native_backend().encode(), np.uint8)}
        lib = S._declare_host()
        # ncclAllReduce(ncclInt64, ncclSum) on a device tensor of widened limbs, through the library's communicator
        limbs = torch.arange(27, dtype=torch.int64, device="cuda") * 0x1_0000_0001
        _lib.check(lib.zk_comm_all_reduce_sum_i64(comm.native(), C.c_void_p(limbs.data_ptr()), 27))
        torch.cuda.synchronize()
        res["limbs"] = limbs.cpu().numpy()
        # torch-level nccl: all-gather of G1 points (the sharded MSM's exchange in the host-driven flow)
        res["points"] = comm.all_gather(points[:2].reshape(-1))
        shard = S.GpuShard.from_array(field, table)
        cs, rp, ch = S.sumcheck_basic_prove_device(comm, shard)
        res.update(basic_claimed=cs, basic_rounds=rp, basic_chal=ch)
        t = zk.Transcript()
        t.append(b"prefix")
        ss = S.GpuSumShard(field, [[zk.MultilinearPolynomial(field, tt) for tt in prod] for prod in sum_tables])
        co, gch, fin = S.sumcheck_gkr_prove_device(comm, ss, claimed, t)
        res.update(gkr_coeffs=co, gkr_chal=gch, gkr_final=fin, gkr_tail=np.frombuffer(t.sample_random_challenge(), np.uint8))
        res["msm"] = S.msm_device(comm, zk.MultilinearPolynomial.vector(0, scalars), zk.G1Bases(points))
        nv = table.shape[0].bit_length() - 1
        res["evaluate"] = S.mle_evaluate(comm, shard.poly, table[:nv])
        rx, ncoll = comm.native_stats()
        res["ncoll"] = np.array([ncoll])
        comm.close()
        np.savez(os.path.join(out_dir, "rank0.npz"), **res)
    finally:
        dist.destroy_process_group()


def run_kzg_open(rank, world, port, backend, table, points, opening, out_dir):
    """sharded open_and_prove: every rank holds the low-bit shard of the table and of the setup's G1 powers"""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as G
        zk = G.import_package()
        from zkmle_amd import _lib
        _lib.check(zk.lib().zk_init(0))
        S = zk.sharded
        comm = S.Comm(device=torch.device("cuda", 0) if backend == "nccl" else None)
        poly = zk.MultilinearPolynomial.vector(0, np.ascontiguousarray(table[rank::world]))
        bases = zk.G1Bases(np.ascontiguousarray(points[rank::world]))
        ev, proofs = S.kzg_open_device(comm, poly, bases, opening)
        commit = S.msm_device(comm, poly, bases)                       # low-bit slices are as good a partition of the terms as any
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ev=ev, proofs=proofs, commit=commit,
                 backend=np.frombuffer(comm.native_backend().encode(), np.uint8))
        comm.close()
    finally:
        dist.destroy_process_group()
