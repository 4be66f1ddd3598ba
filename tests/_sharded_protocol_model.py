"""The sharded protocol driven from the HOST, one exchange at a time -- a step-by-step model of what the C-ABI provers
(`zk_sharded_*`, csrc/zkmle_sharded.hip) do in one call: low-bit shards, per-round all-gather of the partial sums, replicated tail,
whole-table absorb streamed to rank 0, slice-sharded MSM.  TEST INFRASTRUCTURE: it exists so that the exchange logic and the `Comm`
classes of the package run where there is no GPU (per-shard compute through an `engine`: the oracle-backed doubles of
tests/_sharded_workers.py, or the HIP-backed adapters below on a GPU box), and so that the C provers have a second flow to be compared with.
The product never imports it."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as G

zk = G.import_package()
from zkmle_amd import _lib as L                                            # noqa: E402
from zkmle_amd.mle import MultilinearPolynomial, limbs                     # noqa: E402
from zkmle_amd.sharded import _declare_host, fe_add                        # noqa: E402
from zkmle_amd.sumcheck import Transcript, _decl as _sc_decl, lagrange_interpolate   # noqa: E402


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


# ---- per-shard engines on the HIP kernels (single-device entry points of the C ABI) ------------------------------
class HipShard:
    """one rank's local table in HBM; every method launches HIP kernels through the C ABI"""

    def __init__(self, poly):
        self.poly = poly
        self.field = poly.field

    @classmethod
    def from_array(cls, field, arr):
        return cls(MultilinearPolynomial(field, arr))

    def spawn(self, arr):
        return HipShard.from_array(self.field, arr)

    def __len__(self):
        return len(self.poly)

    def half_sums(self):
        return self.poly.half_sums()

    def fold_half_sums(self, r):
        out, sums = self.poly.fold_half_sums(r)
        return HipShard(out), sums

    def fold(self, r):
        return HipShard(MultilinearPolynomial.partial_evaluate(self.poly, 0, r))

    def download(self):
        return self.poly.evaluated_values

    def to_bytes(self):
        return self.poly.convert_to_bytes()


class HipSumShard:
    """one rank's shards of the nprod x nfac tables of a SumPolynomial"""

    def __init__(self, field, tables):
        self.field = field
        self.tables = tables          # list of lists of MultilinearPolynomial
        self.nprod, self.nfac = len(tables), len(tables[0])

    def spawn(self, arrays):
        return HipSumShard(self.field, [[MultilinearPolynomial(self.field, a) for a in prod] for prod in arrays])

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
        return HipSumShard(self.field, outs), ev

    def fold(self, r):
        return HipSumShard(self.field, [[MultilinearPolynomial.partial_evaluate(p, 0, r) for p in prod] for prod in self.tables])

    def download(self):
        return np.stack([np.stack([p.evaluated_values for p in prod]) for prod in self.tables])


def absorb_sharded_table(comm, t, local_bytes, esz, chunk_elems=1 << 15):
    """transcript.append(convert_to_bytes(table)) (prover.rs:38-39) for a low-bit-sharded table: the sponge is sequential,
    so rank 0 alone hashes -- the ranks send their canonical bytes to rank 0 chunk by chunk (gather), rank 0 interleaves a
    chunk into global index order (element j of rank r is global j * G + r) and absorbs it, and the 208-byte sponge state
    is broadcast.  Non-root ranks receive 208 bytes."""
    G = comm.world
    n = len(local_bytes) // esz
    for off in range(0, n, chunk_elems):
        parts = comm.gather_bytes(local_bytes[off * esz:(off + chunk_elems) * esz], 0)
        if parts is not None:
            views = [np.frombuffer(p, np.uint8).reshape(-1, esz) for p in parts]
            t.append(np.stack(views, axis=1).tobytes())
    if G > 1:
        t.import_state(comm.broadcast_u64(t.export_state(), 0))


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
    if absorb_table:
        absorb_sharded_table(comm, t, shard.to_bytes(), esz)
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
