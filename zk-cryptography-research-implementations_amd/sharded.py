"""Multi-GPU (one process per GPU) versions of the path: low-bit table sharding.

GPU g of G = 2^k owns the global indices i == g (mod G) as a contiguous local table
(local index i >> k).  Rounds fold variable 0 = the MSB first (prover.rs:62,
sumcheck_gkr_protocol.rs:57), so partners i and i + N/2 share their low bits: the first
n - k rounds are purely local and the only exchange is a tiny all-gather per round
(2 or 3 field elements per rank) -- RCCL over xGMI when the process group is "nccl", gloo in CPU
tests.  Every rank combines the gathered partial sums and runs the same transcript, so no
broadcast is needed.  The last k rounds run on the gathered G-element table.
The MSM shards by slices: one Pippenger per rank, one all-gather of G affine points, G - 1 additions.

The per-shard compute goes through an `engine` (GpuShard below: HIP kernels via the C ABI).
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .mle import MultilinearPolynomial, limbs
from .sumcheck import Transcript, _decl as _sc_decl, lagrange_interpolate


# ---- host field helpers (control path) ---------------------------------------------------------
def fe_add(field, a, b):
    out = np.zeros(limbs(field), np.uint64)
    L.check(L.lib().zk_fe_add(field, L.p64(np.ascontiguousarray(a, np.uint64)), L.p64(np.ascontiguousarray(b, np.uint64)), L.p64(out)))
    return out


def fe_sum(field, rows):
    acc = np.zeros(limbs(field), np.uint64)
    for r in rows:
        acc = fe_add(field, acc, r)
    return acc


def fe_to_bytes_be(field, a):
    out = np.zeros(8 * limbs(field), np.uint8)
    L.check(L.lib().zk_fe_to_bytes_be(field, L.p64(np.ascontiguousarray(a, np.uint64)), L.p8(out)))
    return out.tobytes()


def fe_to_bytes_le(field, a):
    return fe_to_bytes_be(field, a)[::-1]


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
    lib.zk_rounds_new.argtypes = [C.c_int, C.c_int, L.sz, L.sz, L.sz, L.vp, vpp]
    lib.zk_rounds_free.argtypes = [L.vp]
    lib.zk_rounds_limbs_len.argtypes = [L.vp]
    lib.zk_rounds_limbs_len.restype = C.c_size_t
    lib.zk_rounds_evals.argtypes = [L.vp, vpp, C.c_void_p]
    lib.zk_rounds_fold_evals.argtypes = [L.vp, vpp, vpp, C.c_void_p]
    lib.zk_rounds_absorb.argtypes = [L.vp, C.c_void_p]
    lib.zk_rounds_tail.argtypes = [L.vp, vpp]
    lib.zk_rounds_collect.argtypes = [L.vp, L.vp, u64p, u64p, u64p, u64p]
    for name in ("zk_rounds_new", "zk_rounds_free", "zk_rounds_evals", "zk_rounds_fold_evals", "zk_rounds_absorb", "zk_rounds_tail",
                 "zk_rounds_collect"):
        getattr(lib, name).restype = C.c_int
    lib._sharded_declared = True
    return lib


# ---- communication ---------------------------------------------------------------------------------
class Comm:
    """all-gather of small uint64 arrays over torch.distributed ("nccl" = RCCL on ROCm, or gloo)."""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.device = device
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def all_gather(self, arr):
        import torch
        arr = np.ascontiguousarray(arr, np.uint64)
        if self.world == 1:
            return arr[None].copy()
        t = torch.from_numpy(arr.view(np.int64).copy())
        if self.device is not None:
            t = t.to(self.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        return np.stack([o.cpu().numpy().view(np.uint64) for o in out])

    def all_reduce_sum_(self, t):
        """in-place element-wise integer sum of an int64 tensor (the widened-limb exchange of the device-resident rounds,
        include/zkmle.h zk_rounds): RCCL all-reduce directly on the device tensor, no host synchronisation; a gloo group
        (tests, rehearsal) stages through the host"""
        if self.world == 1:
            return t
        backend = self.dist.get_backend(self.group)
        if backend == "nccl" or not t.is_cuda:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        else:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        return t

    def all_gather_bytes(self, data):
        """equal-length byte strings -> list of bytes, rank order"""
        import torch
        if self.world == 1:
            return [bytes(data)]
        t = torch.frombuffer(bytearray(data), dtype=torch.uint8)
        if self.device is not None:
            t = t.to(self.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        return [bytes(o.cpu().numpy().tobytes()) for o in out]


# ---- per-shard engines -----------------------------------------------------------------------------
class GpuShard:
    """one rank's local table in HBM; every method launches HIP kernels through the C ABI"""

    def __init__(self, poly):
        self.poly = poly
        self.field = poly.field

    @classmethod
    def from_array(cls, field, arr):
        return cls(MultilinearPolynomial(field, arr))

    def spawn(self, arr):
        return GpuShard.from_array(self.field, arr)

    def __len__(self):
        return len(self.poly)

    def half_sums(self):
        return self.poly.half_sums()

    def fold_half_sums(self, r):
        out, sums = self.poly.fold_half_sums(r)
        return GpuShard(out), sums

    def fold(self, r):
        return GpuShard(MultilinearPolynomial.partial_evaluate(self.poly, 0, r))

    def download(self):
        return self.poly.evaluated_values

    def to_bytes(self):
        return self.poly.convert_to_bytes()


def shard_of(global_table, rank, world):
    """the low-bit shard of a host table: elements rank, rank + world, ..."""
    return np.ascontiguousarray(np.asarray(global_table)[rank::world])


# ---- sharded basic sumcheck (prover.rs:35-71) -------------------------------------------------------
def sumcheck_basic_prove(comm, shard, absorb_table=True):
    """-> (claimed_sum, round_polys (n, 2, limbs), challenges (n, limbs)); identical on every rank and
    identical to the single-device proof of the interleaved global table."""
    _declare_host()
    field = shard.field
    G = comm.world
    assert G & (G - 1) == 0, "world size must be a power of two"
    t = Transcript()
    esz = 8 * limbs(field)
    if absorb_table:                                            # prover.rs:38-39, global index order
        parts = comm.all_gather_bytes(shard.to_bytes())
        views = [np.frombuffer(p, np.uint8).reshape(-1, esz) for p in parts]
        t.append(np.stack(views, axis=1).tobytes())             # element j of rank r is global j * G + r
    rounds, chal = [], []
    cur = shard
    replicated = False

    def global_half_sums(engine):
        if replicated:
            return engine.half_sums()
        return combine(comm.all_gather(engine.half_sums()))

    def combine(g):
        return np.stack([fe_sum(field, g[:, 0]), fe_sum(field, g[:, 1])])

    if len(cur) == 1:                                           # fewer local rounds than ranks: go replicated at once
        cur = cur.spawn(comm.all_gather(cur.download()).reshape(G, -1))
        replicated = True
    total_len = len(shard) * G
    if total_len == 1:
        claimed = cur.download()[0]
        t.append(fe_to_bytes_be(field, claimed))
        return claimed, np.zeros((0, 2, limbs(field)), np.uint64), np.zeros((0, limbs(field)), np.uint64)
    sums = global_half_sums(cur)
    claimed = fe_add(field, sums[0], sums[1])                   # prover.rs:28
    t.append(fe_to_bytes_be(field, claimed))                    # :40-41
    nvars = total_len.bit_length() - 1
    for _ in range(nvars):                                      # :46
        rounds.append(sums.copy())
        t.append(fe_to_bytes_be(field, sums[0]) + fe_to_bytes_be(field, sums[1]))    # :52-55
        r = t.random_challenge_as_field_element(field)          # :58
        chal.append(r)
        if len(cur) >= 4:                                       # :61-63 fused with the next round's sums
            cur, local = cur.fold_half_sums(r)
            sums = local if replicated else combine(comm.all_gather(local))
        elif len(cur) == 2:
            cur = cur.fold(r)
            if not replicated and G > 1:                        # one element per rank left: gather and continue replicated
                cur = cur.spawn(comm.all_gather(cur.download()).reshape(G, -1))
                replicated = True
                sums = cur.half_sums()
    return claimed, np.stack(rounds), np.stack(chal)


# ---- sharded GKR sumcheck (sumcheck_gkr_protocol.rs:24-67) -------------------------------------------
class GpuSumShard:
    """one rank's shards of the nprod x nfac tables of a SumPolynomial"""

    def __init__(self, field, tables):
        self.field = field
        self.tables = tables          # list of lists of MultilinearPolynomial
        self.nprod, self.nfac = len(tables), len(tables[0])

    def spawn(self, arrays):
        return GpuSumShard(self.field, [[MultilinearPolynomial(self.field, a) for a in prod] for prod in arrays])

    def __len__(self):
        return len(self.tables[0][0])

    def _arr(self, tabs):
        flat = [p._h for prod in tabs for p in prod]
        return (C.c_void_p * len(flat))(*flat)

    def round_evals(self):
        out = np.zeros((self.nfac + 1, limbs(self.field)), np.uint64)
        L.check(_sc_decl().zk_sumpoly_round_evals(self._arr(self.tables), self.nprod, self.nfac, L.p64(out)))
        return out

    def fold_round_evals(self, r):
        lib = _declare_host()
        half = len(self) // 2
        outs = [[MultilinearPolynomial.alloc(self.field, half) for _ in prod] for prod in self.tables]
        ev = np.zeros((self.nfac + 1, limbs(self.field)), np.uint64)
        L.check(lib.zk_sumpoly_fold_round_evals(self._arr(self.tables), self._arr(outs), self.nprod, self.nfac,
                                                L.p64(np.ascontiguousarray(r, np.uint64)), L.p64(ev)))
        return GpuSumShard(self.field, outs), ev

    def fold(self, r):
        return GpuSumShard(self.field, [[MultilinearPolynomial.partial_evaluate(p, 0, r) for p in prod] for prod in self.tables])

    def download(self):
        return np.stack([np.stack([p.evaluated_values for p in prod]) for prod in self.tables])


def sumcheck_gkr_prove(comm, shard, claimed_sum, transcript):
    """-> (round coefficient rows (n, nfac+1, limbs), challenges (n, limbs)); same bytes as the single-device prover"""
    _declare_host()
    field = shard.field
    G = comm.world
    npts = shard.nfac + 1
    xs = np.stack([_from_u64(field, i) for i in range(npts)])
    transcript.append(fe_to_bytes_be(field, claimed_sum))        # :35
    total_len = len(shard) * G
    nvars = total_len.bit_length() - 1
    cur, replicated = shard, False

    def combine(g):
        return np.stack([fe_sum(field, g[:, k]) for k in range(npts)])

    def gather_tables(engine):
        g = comm.all_gather(engine.download())                   # (G, nprod, nfac, 1, limbs)
        return np.ascontiguousarray(np.transpose(g[:, :, :, 0, :], (1, 2, 0, 3)))

    if nvars == 0:
        return np.zeros((0, npts, limbs(field)), np.uint64), np.zeros((0, limbs(field)), np.uint64)
    if len(cur) == 1:
        cur = cur.spawn(gather_tables(cur))
        replicated = True
    evals = cur.round_evals() if replicated else combine(comm.all_gather(cur.round_evals()))
    coeffs, chal = [], []
    for _ in range(nvars):                                       # :37
        co = lagrange_interpolate(field, xs, evals)              # :49-50
        transcript.append(b"".join(fe_to_bytes_le(field, c) for c in co))   # :52
        coeffs.append(co)
        r = transcript.random_challenge_as_field_element(field)  # :55
        chal.append(r)
        if len(cur) >= 4:
            cur, local = cur.fold_round_evals(r)                 # :57 fused with the next :41
            evals = local if replicated else combine(comm.all_gather(local))
        elif len(cur) == 2:
            cur = cur.fold(r)
            if not replicated and G > 1:
                cur = cur.spawn(gather_tables(cur))
                replicated = True
                evals = cur.round_evals()
    return np.stack(coeffs), np.stack(chal)


def _from_u64(field, v):
    out = np.zeros(limbs(field), np.uint64)
    L.check(L.lib().zk_fe_from_u64(field, v, L.p64(out)))
    return out


# ---- sharded MSM / commit (multilinear_kzg.rs:37-42) --------------------------------------------------
def g1_sum(points):
    lib = _declare_host()
    acc = np.zeros(12, np.uint64)
    for p in points:
        out = np.zeros(12, np.uint64)
        L.check(lib.zk_g1_add(L.p64(acc), L.p64(np.ascontiguousarray(p, np.uint64)), L.p64(out)))
        acc = out
    return acc


def msm(comm, local_msm):
    """local_msm: () -> this rank's partial point (12 limbs).  One all-gather of G points, G - 1 additions."""
    return g1_sum(comm.all_gather(local_msm()))


# ---- device-resident sharded provers: no host round trip per round (include/zkmle.h, zk_rounds) -----------------------
class DeviceRounds:
    """zk_rounds handle + the int64 limb buffer the ranks all-reduce.  One per sumcheck."""

    def __init__(self, field, mode, nprod, nfac, nrounds, transcript, device=None):
        import torch
        lib = _declare_host()
        self.field, self.mode, self.nprod, self.nfac, self.nrounds = field, mode, nprod, nfac, nrounds
        self._t = transcript
        h = C.c_void_p()
        L.check(lib.zk_rounds_new(field, mode, nprod, nfac, nrounds, transcript._h, C.byref(h)))
        self._h = h
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.limbs = torch.zeros(int(lib.zk_rounds_limbs_len(h)), dtype=torch.int64, device=dev)

    def __del__(self):
        if getattr(self, "_h", None) is not None and L._lib is not None:
            L.lib().zk_rounds_free(self._h)
            self._h = None

    @staticmethod
    def _arr(tabs):
        flat = [p._h for p in tabs]
        return (C.c_void_p * len(flat))(*flat)

    def evals(self, tabs):
        L.check(L.lib().zk_rounds_evals(self._h, self._arr(tabs), C.c_void_p(self.limbs.data_ptr())))

    def fold_evals(self, tabs):
        half = len(tabs[0]) // 2
        outs = [MultilinearPolynomial.alloc(self.field, half) for _ in tabs]
        L.check(L.lib().zk_rounds_fold_evals(self._h, self._arr(tabs), self._arr(outs), C.c_void_p(self.limbs.data_ptr())))
        return outs

    def absorb(self):
        L.check(L.lib().zk_rounds_absorb(self._h, C.c_void_p(self.limbs.data_ptr())))

    def tail(self, tabs):
        L.check(L.lib().zk_rounds_tail(self._h, self._arr(tabs)))

    def collect(self, want_final):
        n = limbs(self.field)
        npts = self.nfac + 1
        claimed = np.zeros(n, np.uint64)
        msgs = np.zeros((self.nrounds, npts, n), np.uint64)
        chal = np.zeros((self.nrounds, n), np.uint64)
        fin = np.zeros((self.nprod * self.nfac, n), np.uint64)
        L.check(L.lib().zk_rounds_collect(self._h, self._t._h, L.p64(claimed), L.p64(msgs), L.p64(chal), L.p64(fin) if want_final else None))
        return claimed, msgs, chal, fin


def _device_rounds(comm, field, mode, tabs, nprod, nfac, transcript):
    """shared driver: tabs = this rank's local tables (MultilinearPolynomial handles, low-bit shards).
    -> (claimed, messages, challenges, final values (ntab, limbs))"""
    G = comm.world
    assert G & (G - 1) == 0, "world size must be a power of two"
    nloc = len(tabs[0]).bit_length() - 1
    g = G.bit_length() - 1
    nrounds = nloc + g
    dr = DeviceRounds(field, mode, nprod, nfac, nrounds, transcript, comm.device)
    cur = tabs
    if nloc >= 1:
        dr.evals(cur)
        comm.all_reduce_sum_(dr.limbs)                          # the round's only exchange, on the device
        dr.absorb()
        while len(cur[0]) >= 4:
            cur = dr.fold_evals(cur)
            comm.all_reduce_sum_(dr.limbs)
            dr.absorb()
        cur = dr.fold_evals(cur)                                # 2 entries -> 1
    if g == 0:
        claimed, msgs, chal, _ = dr.collect(False)
        fin = np.stack([p.evaluated_values[0] for p in cur])
        return claimed, msgs, chal, fin
    # one entry per rank and table left: gather them (global index = rank) and finish replicated, in one launch
    mine = np.stack([p.evaluated_values[0] for p in cur])       # (ntab, limbs)
    allv = comm.all_gather(mine)                                # (G, ntab, limbs)
    rep = [MultilinearPolynomial(field, np.ascontiguousarray(allv[:, k, :])) for k in range(len(cur))]
    dr.evals(rep)                                               # already global: no all-reduce
    dr.absorb()
    dr.tail(rep)
    return dr.collect(True)


def sumcheck_gkr_prove_device(comm, shard, claimed_sum, transcript):
    """sumcheck_gkr_prove with the transcript on the device: same bytes, one all-reduce per local round, one host
    synchronisation at the gather and one at the end.  -> (coefficient rows, challenges, final table values)"""
    field = shard.field
    transcript.append(fe_to_bytes_be(field, claimed_sum))       # sumcheck_gkr_protocol.rs:35
    flat = [p for prod in shard.tables for p in prod]
    if len(flat[0]) * comm.world == 1:
        return (np.zeros((0, shard.nfac + 1, limbs(field)), np.uint64), np.zeros((0, limbs(field)), np.uint64),
                np.stack([p.evaluated_values[0] for p in flat]))
    _, msgs, chal, fin = _device_rounds(comm, field, 1, flat, shard.nprod, shard.nfac, transcript)
    return msgs, chal, fin


def sumcheck_basic_prove_device(comm, shard, absorb_table=True):
    """sumcheck_basic_prove with device-resident rounds.  -> (claimed_sum, round_polys, challenges)"""
    _declare_host()
    field = shard.field
    G = comm.world
    t = Transcript()
    esz = 8 * limbs(field)
    if absorb_table:                                            # prover.rs:38-39, global index order
        parts = comm.all_gather_bytes(shard.to_bytes())
        views = [np.frombuffer(p, np.uint8).reshape(-1, esz) for p in parts]
        t.append(np.stack(views, axis=1).tobytes())
    if len(shard) * G == 1:
        claimed = shard.download()[0]
        t.append(fe_to_bytes_be(field, claimed))
        return claimed, np.zeros((0, 2, limbs(field)), np.uint64), np.zeros((0, limbs(field)), np.uint64)
    claimed, msgs, chal, _ = _device_rounds(comm, field, 0, [shard.poly], 1, 1, t)
    return claimed, msgs, chal
