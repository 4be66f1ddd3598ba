"""Multi-GPU (one process per GPU) versions of the path: low-bit table sharding.

GPU g of G = 2^k owns the global indices i == g (mod G) as a contiguous local table
(local index i >> k).  Rounds fold variable 0 = the MSB first (prover.rs:62,
sumcheck_gkr_protocol.rs:57), so partners i and i + N/2 share their low bits: the first
n - k rounds are purely local and the only exchange is a tiny all-gather per round
(2 or 3 field elements per rank) -- RCCL over xGMI when the process group is "nccl", gloo in CPU
tests.  Every rank combines the gathered partial sums and runs the same transcript, so no
broadcast is needed.  The last k rounds run on the gathered G-element table.
The MSM shards by slices: one Pippenger per rank, one all-gather of G affine points, G - 1 additions.

This module is the ctypes face of that: `*_prove_device`, `mle_evaluate`, `msm_device`, `kzg_open_device` are ONE call each into
the C ABI (include/zkmle.h `zk_sharded_*`, csrc/zkmle_sharded.hip), where the rounds, the RCCL all-reduce of the limb sums and the
transcript kernel are enqueued back to back on the prover's stream; `Comm.native()` hands the library its communicator (an RCCL
communicator of its own when the process group is "nccl", exchange callbacks over the group otherwise), `LocalGroup` runs the ranks as
threads of one process.  (A host-driven, exchange-by-exchange model of the same protocol lives with the tests:
tests/_sharded_protocol_model.py.)
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .mle import MultilinearPolynomial, limbs


# ---- host field helpers (control path) ---------------------------------------------------------
def fe_add(field, a, b):
    out = np.zeros(limbs(field), np.uint64)
    L.check(L.lib().zk_fe_add(field, L.p64(np.ascontiguousarray(a, np.uint64)), L.p64(np.ascontiguousarray(b, np.uint64)), L.p64(out)))
    return out


class HostOps(C.Structure):
    """zk_comm_host_ops (include/zkmle.h): exchange callbacks on HOST memory"""
    _fields_ = [("ctx", C.c_void_p),
                ("all_reduce_sum_i64", C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_size_t)),
                ("all_gather", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)),
                ("gather", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)),
                ("broadcast", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int))]


def _declare_host():
    lib = L.lib()
    if getattr(lib, "_sharded_declared", False):
        return lib
    u64p = L.u64p
    for name in ("zk_fe_add", "zk_fe_sub", "zk_fe_mul"):
        getattr(lib, name).argtypes = [C.c_int, u64p, u64p, u64p]
        getattr(lib, name).restype = C.c_int
    lib.zk_g1_add.argtypes = [u64p, u64p, u64p]
    lib.zk_g1_add.restype = C.c_int
    lib.zk_sumpoly_fold_round_evals.argtypes = [C.POINTER(L.vp), C.POINTER(L.vp), L.sz, L.sz, u64p, u64p]
    lib.zk_sumpoly_fold_round_evals.restype = C.c_int
    vpp = C.POINTER(L.vp)
    u8p, i64p = L.u8p, C.POINTER(C.c_int64)
    lib.zk_comm_unique_id.argtypes = [u8p]
    lib.zk_comm_init_rccl.argtypes = [u8p, C.c_int, C.c_int, vpp]
    lib.zk_comm_from_host_ops.argtypes = [C.POINTER(HostOps), C.c_int, C.c_int, vpp]
    lib.zk_comm_free.argtypes = [L.vp]
    lib.zk_comm_local_group_new.argtypes = [C.c_int, vpp]
    lib.zk_comm_local_group_free.argtypes = [L.vp]
    lib.zk_comm_local_group_abort.argtypes = [L.vp]
    lib.zk_comm_from_local_group.argtypes = [L.vp, C.c_int, vpp]
    lib.zk_comm_backend.argtypes = [L.vp]
    lib.zk_comm_backend.restype = C.c_char_p
    lib.zk_comm_stats.argtypes = [L.vp, u64p, u64p]
    lib.zk_comm_all_reduce_sum_i64.argtypes = [L.vp, L.vp, L.sz]
    lib.zk_comm_all_gather.argtypes = [L.vp, L.vp, L.vp, L.sz]
    lib.zk_comm_broadcast.argtypes = [L.vp, L.vp, L.sz, C.c_int]
    lib.zk_comm_host_exchange.argtypes = [L.vp, C.c_int, L.vp, L.vp, L.sz, C.c_int]
    lib.zk_comm_host_exchange.restype = C.c_int
    lib.zk_sharded_sumcheck_basic_prove.argtypes = [L.vp, L.vp, C.c_int, u64p, u64p, u64p]
    lib.zk_sharded_sumcheck_gkr_prove.argtypes = [L.vp, vpp, L.sz, L.sz, u64p, L.vp, u64p, u64p, u64p]
    lib.zk_sharded_mle_evaluate.argtypes = [L.vp, L.vp, u64p, L.sz, u64p]
    lib.zk_sharded_msm_g1.argtypes = [L.vp, L.vp, L.vp, C.c_int, u64p, L.vp]
    lib.zk_sharded_kzg_open.argtypes = [L.vp, L.vp, L.vp, L.vp, u64p, L.sz, u64p, u64p]
    lib.zk_sharded_kzg_open.restype = C.c_int
    for name in ("zk_comm_unique_id", "zk_comm_init_rccl", "zk_comm_from_host_ops", "zk_comm_free", "zk_comm_stats",
                 "zk_comm_local_group_new", "zk_comm_local_group_free", "zk_comm_local_group_abort", "zk_comm_from_local_group",
                 "zk_comm_all_reduce_sum_i64", "zk_comm_all_gather", "zk_comm_broadcast", "zk_sharded_sumcheck_basic_prove",
                 "zk_sharded_sumcheck_gkr_prove", "zk_sharded_mle_evaluate", "zk_sharded_msm_g1"):
        getattr(lib, name).restype = C.c_int
    lib._sharded_declared = True
    return lib


# ---- communication ---------------------------------------------------------------------------------
class Comm:
    """This rank's end of the exchange.  Python-level collectives on small uint64 arrays over torch.distributed ("nccl" =
    RCCL on ROCm, or gloo) for the host-driven flow, and `native()`: the zk_comm the C-ABI provers run their collectives
    on.  `bytes_received` counts the payload this rank received through the Python-level collectives."""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.device = device
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bytes_received = 0
        self._native = None
        self._keep = None
        self._rccl_failed = False
        self.native_note = None

    def backend(self):
        return self.dist.get_backend(self.group) if self.dist.is_initialized() else "none"

    def _dev(self, t):
        return t.to(self.device) if self.device is not None else t

    def all_gather(self, arr):
        import torch
        arr = np.ascontiguousarray(arr, np.uint64)
        if self.world == 1:
            return arr[None].copy()
        t = self._dev(torch.from_numpy(arr.view(np.int64).copy()))
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        self.bytes_received += (self.world - 1) * arr.nbytes
        return np.stack([o.cpu().numpy().view(np.uint64) for o in out])

    def gather_bytes(self, data, root=0):
        """equal-length byte strings -> list of bytes in rank order on `root`, None elsewhere"""
        import torch
        if self.world == 1:
            return [bytes(data)]
        t = self._dev(torch.frombuffer(bytearray(data), dtype=torch.uint8))
        out = [torch.empty_like(t) for _ in range(self.world)] if self.rank == root else None
        self.dist.gather(t, out, dst=root, group=self.group)
        if self.rank != root:
            return None
        self.bytes_received += (self.world - 1) * len(data)
        return [bytes(o.cpu().numpy().tobytes()) for o in out]

    def broadcast_u64(self, arr, root=0):
        """arr (uint64, same shape on every rank) <- root's arr"""
        import torch
        arr = np.ascontiguousarray(arr, np.uint64)
        if self.world == 1:
            return arr
        t = self._dev(torch.from_numpy(arr.view(np.int64).copy()))
        self.dist.broadcast(t, src=root, group=self.group)
        if self.rank != root:
            self.bytes_received += arr.nbytes
        return t.cpu().numpy().view(np.uint64)

    # ---- the communicator of the C-ABI provers (csrc/zkmle_sharded.hip) ----
    def native(self):
        """zk_comm handle: an RCCL communicator created by the library itself (ncclCommInitRank on this process's device,
        the unique id travels over the torch process group) when the group's backend is "nccl"; a one-rank communicator
        without any transport when there is no process group; exchange callbacks over the group (host memory; gloo)
        otherwise."""
        if self._native is not None:
            return self._native
        import torch
        lib = _declare_host()
        h = C.c_void_p()
        backend = self.backend()
        if backend == "none":
            ops = HostOps()
            L.check(lib.zk_comm_from_host_ops(C.byref(ops), 1, 0, C.byref(h)))
        elif backend == "nccl" and not self._rccl_failed:
            dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
            uid = np.zeros(128, np.uint8)
            try:                                                    # every rank reaches the broadcast and the vote below, whatever fails
                if self.rank == 0:
                    L.check(lib.zk_comm_unique_id(L.p8(uid)))
            except Exception as e:                                  # noqa: BLE001
                self.native_note = f"zk_comm_unique_id failed ({e!r})"[:300]
                self._rccl_failed = True
            if self.world > 1:
                t = torch.from_numpy(uid).to(dev)
                self.dist.broadcast(t, src=0, group=self.group)
                uid = t.cpu().numpy()
            if not uid.any():
                self._rccl_failed = True
            if not self._rccl_failed:
                try:
                    L.check(lib.zk_comm_init_rccl(L.p8(np.ascontiguousarray(uid)), self.world, self.rank, C.byref(h)))
                except Exception as e:                              # noqa: BLE001
                    # the library could not create its RCCL communicator: the ranks agree on it (vote below) and carry the exchange
                    # over a gloo group created for the purpose (host staged); native_note says so
                    self.native_note = f"zk_comm_init_rccl failed ({e!r}); exchange callbacks over a gloo group instead"[:300]
                    self._rccl_failed = True
            flag = torch.tensor([1 if self._rccl_failed else 0], device=self.device if self.device is not None else "cuda")
            if self.world > 1:
                self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX, group=self.group)
            if int(flag.item()):
                if h.value:
                    lib.zk_comm_free(h)
                self._rccl_failed = True
                self.native_note = self.native_note or "another rank could not create its RCCL communicator; exchange callbacks over gloo"
                self.group = self.dist.new_group(backend="gloo") if self.world > 1 else None
                self.device = None
                return self.native()
        else:
            dist, group, world, rank = self.dist, self.group, self.world, self.rank
            if self._rccl_failed and world == 1:
                ops = HostOps()
                L.check(lib.zk_comm_from_host_ops(C.byref(ops), 1, 0, C.byref(h)))
                self._native = h
                return h

            def view(ptr, nbytes):
                return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (nbytes,)))

            def all_reduce(ctx, buf, count):
                try:
                    t = torch.from_numpy(np.ctypeslib.as_array(buf, (count,)))
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                    return 0
                except Exception:                                  # noqa: BLE001  (must not unwind through C)
                    return 1

            def all_gather(ctx, send, recv, nbytes):
                try:
                    out = view(recv, nbytes * world)
                    dist.all_gather(list(out.view(world, nbytes).unbind(0)), view(send, nbytes).clone(), group=group)
                    return 0
                except Exception:                                  # noqa: BLE001
                    return 1

            def gather(ctx, send, recv, nbytes, root):
                try:
                    outs = list(view(recv, nbytes * world).view(world, nbytes).unbind(0)) if rank == root else None
                    dist.gather(view(send, nbytes).clone(), outs, dst=root, group=group)
                    return 0
                except Exception:                                  # noqa: BLE001
                    return 1

            def broadcast(ctx, buf, nbytes, root):
                try:
                    dist.broadcast(view(buf, nbytes), src=root, group=group)
                    return 0
                except Exception:                                  # noqa: BLE001
                    return 1

            ops = HostOps()
            ops.ctx = None
            ops.all_reduce_sum_i64 = HostOps._fields_[1][1](all_reduce)
            ops.all_gather = HostOps._fields_[2][1](all_gather)
            ops.gather = HostOps._fields_[3][1](gather)
            ops.broadcast = HostOps._fields_[4][1](broadcast)
            self._keep = ops                                        # the callbacks live as long as the communicator
            L.check(lib.zk_comm_from_host_ops(C.byref(ops), world, rank, C.byref(h)))
        self._native = h
        return h

    def native_backend(self):
        return _declare_host().zk_comm_backend(self.native()).decode()

    def native_stats(self):
        """(payload bytes received, collectives issued) by the C-ABI provers on this rank"""
        rx, n = C.c_uint64(0), C.c_uint64(0)
        L.check(_declare_host().zk_comm_stats(self.native(), C.byref(rx), C.byref(n)))
        return int(rx.value), int(n.value)

    def close(self):
        if self._native is not None and L._lib is not None:
            L.lib().zk_comm_free(self._native)
        self._native = None

    def __del__(self):
        try:
            self.close()
        except Exception:                                          # noqa: BLE001
            pass


class LocalGroup:
    """The ranks as THREADS of this process (include/zkmle.h zk_comm_local_group_*): what a one-process host with one thread per GPU
    uses, and how more ranks than a one-GPU box allows processes (8-way config 5) are rehearsed on one device.  `comm(rank)` is
    called by rank's thread and gives the same interface as Comm; `abort()` wakes every rank waiting in an exchange."""

    def __init__(self, world):
        import threading
        lib = _declare_host()
        self.world = world
        self._h = C.c_void_p()
        L.check(lib.zk_comm_local_group_new(world, C.byref(self._h)))
        self._bar = threading.Barrier(world)

    def comm(self, rank):
        return LocalComm(self, rank)

    def abort(self):
        _declare_host().zk_comm_local_group_abort(self._h)
        self._bar.abort()

    def close(self):
        if self._h is not None and L._lib is not None:
            L.lib().zk_comm_local_group_free(self._h)
        self._h = None


class LocalComm:
    """one thread's end of a LocalGroup; same surface as Comm for the product path (`native()`, stats, small host all-gathers)"""

    def __init__(self, group, rank):
        self.group_, self.rank, self.world = group, rank, group.world
        self.bytes_received = 0
        self.native_note = None
        self._native = C.c_void_p()
        L.check(_declare_host().zk_comm_from_local_group(group._h, rank, C.byref(self._native)))

    def backend(self):
        return "local-threads"

    def native(self):
        return self._native

    def native_backend(self):
        return _declare_host().zk_comm_backend(self._native).decode()

    def native_stats(self):
        rx, n = C.c_uint64(0), C.c_uint64(0)
        L.check(_declare_host().zk_comm_stats(self._native, C.byref(rx), C.byref(n)))
        return int(rx.value), int(n.value)

    def barrier(self):
        self.group_._bar.wait()

    def _xchg(self, op, buf, recv, n, root=0):
        L.check(_declare_host().zk_comm_host_exchange(self._native, op, buf.ctypes.data_as(C.c_void_p),
                                                      recv.ctypes.data_as(C.c_void_p) if recv is not None else None, n, root))

    def all_gather(self, arr):
        arr = np.ascontiguousarray(arr, np.uint64)
        out = np.zeros((self.world,) + arr.shape, np.uint64)
        self._xchg(1, arr, out, arr.nbytes)
        self.bytes_received += (self.world - 1) * arr.nbytes
        return out

    def all_reduce_sum_i64(self, arr):
        buf = np.ascontiguousarray(arr, np.int64).copy()
        self._xchg(0, buf, None, buf.size)
        return buf

    def gather_bytes(self, data, root=0):
        send = np.frombuffer(bytes(data), np.uint8).copy()
        recv = np.zeros((self.world, len(send)), np.uint8) if self.rank == root else None
        self._xchg(2, send, recv, len(send), root)
        if self.rank != root:
            return None
        self.bytes_received += (self.world - 1) * len(send)
        return [r.tobytes() for r in recv]

    def broadcast_u64(self, arr, root=0):
        buf = np.ascontiguousarray(arr, np.uint64).copy()
        self._xchg(3, buf, None, buf.nbytes, root)
        if self.rank != root:
            self.bytes_received += buf.nbytes
        return buf

    def close(self):
        if self._native is not None and L._lib is not None:
            L.lib().zk_comm_free(self._native)
        self._native = None


# ---- one rank's tables -------------------------------------------------------------------------------
class GpuShard:
    """one rank's local table in HBM (the low-bit shard of the global table)"""

    def __init__(self, poly):
        self.poly = poly
        self.field = poly.field

    @classmethod
    def from_array(cls, field, arr):
        return cls(MultilinearPolynomial(field, arr))

    def __len__(self):
        return len(self.poly)


class GpuSumShard:
    """one rank's shards of the nprod x nfac tables of a SumPolynomial"""

    def __init__(self, field, tables):
        self.field = field
        self.tables = tables          # list of lists of MultilinearPolynomial
        self.nprod, self.nfac = len(tables), len(tables[0])

    def __len__(self):
        return len(self.tables[0][0])


def shard_of(global_table, rank, world):
    """the low-bit shard of a host table: elements rank, rank + world, ..."""
    return np.ascontiguousarray(np.asarray(global_table)[rank::world])


# ---- the product path: one C-ABI call per prover (include/zkmle.h zk_sharded_*, csrc/zkmle_sharded.hip) -------------------
def _tab_arr(tabs):
    return (C.c_void_p * len(tabs))(*[p._h for p in tabs])


def sumcheck_gkr_prove_device(comm, shard, claimed_sum, transcript):
    """sumcheck_gkr_protocol::prove (:24-67) on this rank's low-bit shards: per local round one fused kernel, ONE all-reduce
    of 27 int64 words on the prover's stream (RCCL) and the transcript kernel; the <= 2048 last global entries are gathered
    and finished replicated in one launch.  Same bytes as the single-device prover on every rank.
    -> (coefficient rows, challenges, final table values)"""
    lib = _declare_host()
    field = shard.field
    flat = [p for prod in shard.tables for p in prod]
    n = limbs(field)
    nrounds = (len(flat[0]) * comm.world).bit_length() - 1
    co = np.zeros((nrounds, shard.nfac + 1, n), np.uint64)
    ch = np.zeros((nrounds, n), np.uint64)
    fin = np.zeros((len(flat), n), np.uint64)
    L.check(lib.zk_sharded_sumcheck_gkr_prove(comm.native(), _tab_arr(flat), shard.nprod, shard.nfac,
                                              L.p64(np.ascontiguousarray(claimed_sum, np.uint64)), transcript._h,
                                              L.p64(co), L.p64(ch), L.p64(fin)))
    return co, ch, fin


def sumcheck_basic_prove_device(comm, shard, absorb_table=True):
    """Prover::prove (prover.rs:35-71) of the sharded table through the C ABI.  -> (claimed_sum, round_polys, challenges)"""
    lib = _declare_host()
    field = shard.field
    n = limbs(field)
    nrounds = (len(shard) * comm.world).bit_length() - 1
    claimed = np.zeros(n, np.uint64)
    rp = np.zeros((nrounds, 2, n), np.uint64)
    ch = np.zeros((nrounds, n), np.uint64)
    L.check(lib.zk_sharded_sumcheck_basic_prove(comm.native(), shard.poly._h, 1 if absorb_table else 0, L.p64(claimed),
                                                L.p64(rp), L.p64(ch)))
    return claimed, rp, ch


def mle_evaluate(comm, poly, values):
    """MultilinearPolynomial::evaluate (evaluation_form.rs:21-33) of the sharded table at log2(local) + log2(G) points"""
    lib = _declare_host()
    v = np.ascontiguousarray(values, np.uint64).reshape(-1, limbs(poly.field))
    out = np.zeros(limbs(poly.field), np.uint64)
    L.check(lib.zk_sharded_mle_evaluate(comm.native(), poly._h, L.p64(v), v.shape[0], L.p64(out)))
    return out


def msm_device(comm, scalars, bases, window_bits=0, with_stats=False):
    """commit_to_polynomial (multilinear_kzg.rs:37-42) with the terms sliced over the ranks, through the C ABI"""
    from .kzg import MsmStats
    lib = _declare_host()
    out = np.zeros(12, np.uint64)
    st = MsmStats()
    L.check(lib.zk_sharded_msm_g1(comm.native(), scalars._h, bases._h, window_bits, L.p64(out), C.byref(st)))
    return (out, st.as_dict()) if with_stats else out


def kzg_open_device(comm, poly, bases_local, opening, key_local=None):
    """open_and_prove (multilinear_kzg.rs:50-126) of the sharded table: poly = this rank's low-bit shard, bases_local = the same shard of
    the setup's G1 powers.  -> (evaluation, proofs (m + k, 12)), the single-device opening on every rank"""
    lib = _declare_host()
    op = np.ascontiguousarray(opening, np.uint64).reshape(-1, 4)
    ev = np.zeros(4, np.uint64)
    proofs = np.zeros((op.shape[0], 12), np.uint64)
    L.check(lib.zk_sharded_kzg_open(comm.native(), poly._h, bases_local._h, key_local._h if key_local is not None else None,
                                    L.p64(op), op.shape[0], L.p64(ev), L.p64(proofs)))
    return ev, proofs
