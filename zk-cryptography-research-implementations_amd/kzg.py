"""Host mirror of multilinear_kzg/src/{multilinear_kzg,trusted_setup}.rs (prover side) over the C ABI."""
import ctypes as C

import numpy as np

from . import _lib as L
from .mle import MultilinearPolynomial


class MsmStats(C.Structure):
    _fields_ = [("window_bits", C.c_int), ("windows", C.c_int), ("terms", C.c_uint64), ("entries", C.c_uint64),
                ("segments", C.c_uint64), ("ms_digits", C.c_float), ("ms_sort", C.c_float), ("ms_buckets", C.c_float),
                ("ms_reduce", C.c_float), ("ms_total", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def _decl():
    lib = L.lib()
    if getattr(lib, "_kzg_declared", False):
        return lib
    vp, sz, u64p = L.vp, L.sz, L.u64p
    lib.zk_g1_bases_len.restype = sz
    lib.zk_g1_bases_len.argtypes = [vp]
    sigs = {
        "zk_g1_bases_upload": [u64p, sz, C.POINTER(vp)], "zk_g1_bases_download": [vp, u64p], "zk_g1_bases_free": [vp],
        "zk_g1_bases_synthetic": [sz, u64p, u64p, C.POINTER(vp)], "zk_g1_generator": [u64p], "zk_g1_is_on_curve": [u64p],
        "zk_msm_g1": [vp, vp, C.c_int, u64p, C.POINTER(MsmStats)],
        "zk_g1_bases_precompute": [vp, C.c_int], "zk_g1_bases_precomputed_window": [vp],
        "zk_kzg_lagrange_basis": [u64p, sz, C.POINTER(vp)], "zk_kzg_setup_g1": [u64p, sz, C.POINTER(vp)],
        "zk_kzg_commit": [vp, vp, u64p],
        "zk_kzg_opening_key_new": [vp, C.POINTER(vp)], "zk_kzg_opening_key_free": [vp],
        "zk_kzg_opening_key_precompute": [vp, C.c_int, C.c_size_t],
        "zk_kzg_open": [vp, vp, vp, u64p, sz, sz, u64p, u64p],
        "zk_g2_generator": [u64p], "zk_g2_is_on_curve": [u64p], "zk_g2_add": [u64p, u64p, u64p], "zk_g2_mul_fr": [u64p, u64p, u64p],
        "zk_pairing": [u64p, u64p, u64p], "zk_pairing_product_is_one": [u64p, u64p, sz, C.POINTER(C.c_int)],
        "zk_kzg_setup_g2": [u64p, sz, u64p],
        "zk_kzg_verify": [u64p, u64p, sz, u64p, u64p, sz, u64p, sz, C.POINTER(C.c_int)],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib._kzg_declared = True
    return lib


def g1_generator():
    out = np.zeros(12, np.uint64)
    L.check(_decl().zk_g1_generator(L.p64(out)))
    return out


def g1_is_on_curve(p):
    return bool(_decl().zk_g1_is_on_curve(L.p64(np.ascontiguousarray(p, np.uint64))))


class G1Bases:
    """HBM-resident affine G1 points (`TrustedSetup.g1_powers_of_tau`, trusted_setup.rs:5-8)."""

    def __init__(self, affine=None, _handle=None):
        if _handle is not None:
            self._h = _handle
            return
        pts = np.ascontiguousarray(affine, np.uint64).reshape(-1, 12)
        h = C.c_void_p()
        L.check(_decl().zk_g1_bases_upload(L.p64(pts), pts.shape[0], C.byref(h)))
        self._h = h

    @classmethod
    def synthetic(cls, n, a, d):
        h = C.c_void_p()
        L.check(_decl().zk_g1_bases_synthetic(n, L.p64(np.ascontiguousarray(a, np.uint64)),
                                              L.p64(np.ascontiguousarray(d, np.uint64)), C.byref(h)))
        return cls(_handle=h)

    def precompute(self, window_bits=0):
        """one copy of the points per window (zk_g1_bases_precompute): later MSMs / commits on these bases use ONE bucket set"""
        L.check(_decl().zk_g1_bases_precompute(self._h, window_bits))
        return _decl().zk_g1_bases_precomputed_window(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                L.lib().zk_g1_bases_free(self._h)
            except Exception:
                pass
            self._h = None

    def __len__(self):
        return _decl().zk_g1_bases_len(self._h)

    def points(self):
        out = np.zeros((len(self), 12), np.uint64)
        L.check(_decl().zk_g1_bases_download(self._h, L.p64(out)))
        return out


def msm(scalars, bases, window_bits=0, with_stats=False):
    """sum_i [s_i] B_i -> affine point (12 limbs)"""
    out = np.zeros(12, np.uint64)
    st = MsmStats()
    L.check(_decl().zk_msm_g1(scalars._h, bases._h, window_bits, L.p64(out), C.byref(st)))
    return (out, st.as_dict()) if with_stats else out


def compute_lagrange_basis(taus):
    """trusted_setup.rs:24-49 -> MultilinearPolynomial over Fr (the eq table)"""
    t = np.ascontiguousarray(taus, np.uint64).reshape(-1, 4)
    h = C.c_void_p()
    L.check(_decl().zk_kzg_lagrange_basis(L.p64(t), t.shape[0], C.byref(h)))
    return MultilinearPolynomial(L.FR381, _handle=h)


class TrustedSetup:
    """trusted_setup.rs:5-22.  g1_powers_of_tau live in HBM (a G1Bases handle); g2_powers_of_tau ([tau_i] G2, n x 24 limbs,
    :62-72) are verifier-side host data, computed on first use."""

    def __init__(self, g1_powers_of_tau, n_g2, taus=None):
        self.g1_powers_of_tau = g1_powers_of_tau
        self.n_g2_powers_of_tau = n_g2
        self._taus = taus
        self._g2 = None
        self._opening_key = None

    @classmethod
    def initialize_setup(cls, taus):
        t = np.ascontiguousarray(taus, np.uint64).reshape(-1, 4)
        h = C.c_void_p()
        L.check(_decl().zk_kzg_setup_g1(L.p64(t), t.shape[0], C.byref(h)))
        return cls(G1Bases(_handle=h), t.shape[0], t.copy())

    def precompute_for_commits(self, window_bits=0):
        """optional, once per setup: window-shifted copies of the G1 powers (zk_g1_bases_precompute); later commits use one bucket set"""
        return self.g1_powers_of_tau.precompute(window_bits)

    def precompute_for_opens(self, window_bits=0, min_points=0):
        """optional, once per setup: the same for the pre-summed levels of the opening key (zk_kzg_opening_key_precompute): the level MSMs of
        open_and_prove then use one bucket set each"""
        L.check(_decl().zk_kzg_opening_key_precompute(self.opening_key(), window_bits, min_points))

    @property
    def g2_powers_of_tau(self):
        if self._g2 is None:
            if self._taus is None:
                raise L.ZkError(L.ZK_E_ARG, "this setup was built without the taus: no G2 powers")
            out = np.zeros((self._taus.shape[0], 24), np.uint64)
            L.check(_decl().zk_kzg_setup_g2(L.p64(self._taus), self._taus.shape[0], L.p64(out)))
            self._g2 = out
        return self._g2

    def opening_key(self):
        if self._opening_key is None:
            h = C.c_void_p()
            L.check(_decl().zk_kzg_opening_key_new(self.g1_powers_of_tau._h, C.byref(h)))
            self._opening_key = h
        return self._opening_key

    def __del__(self):
        if getattr(self, "_opening_key", None):
            try:
                L.lib().zk_kzg_opening_key_free(self._opening_key)
            except Exception:
                pass
            self._opening_key = None


class MultilinearKZGProof:                       # multilinear_kzg.rs:16-20
    def __init__(self, evaluation, proofs):
        self.evaluation = evaluation
        self.proofs = proofs


class MultilinearKZG:
    @staticmethod
    def commit_to_polynomial(polynomial, trusted_setup):       # :25-45
        out = np.zeros(12, np.uint64)
        L.check(_decl().zk_kzg_commit(polynomial._h, trusted_setup.g1_powers_of_tau._h, L.p64(out)))
        return out

    @staticmethod
    def open_and_prove(polynomial, trusted_setup, opening_values):   # :50-126
        o = np.ascontiguousarray(opening_values, np.uint64).reshape(-1, 4)
        ev = np.zeros(4, np.uint64)
        proofs = np.zeros((max(o.shape[0], 1), 12), np.uint64)
        nvars = polynomial.number_of_variables()
        key = trusted_setup.opening_key() if (o.shape[0] == nvars == trusted_setup.n_g2_powers_of_tau and
                                              len(trusted_setup.g1_powers_of_tau) == len(polynomial)) else None
        optr = L.p64(o) if o.size else L.p64(ev)
        L.check(_decl().zk_kzg_open(polynomial._h, trusted_setup.g1_powers_of_tau._h, key, optr, o.shape[0],
                                    trusted_setup.n_g2_powers_of_tau, L.p64(ev), L.p64(proofs)))
        return MultilinearKZGProof(ev, proofs[: o.shape[0]])

    @staticmethod
    def verify(trusted_setup, commitment, opening_values, proof):    # :131-158 (pairings on the host, csrc/pairing.h)
        o = np.ascontiguousarray(opening_values, np.uint64).reshape(-1, 4)
        prs = np.ascontiguousarray(proof.proofs, np.uint64).reshape(-1, 12)
        g2 = trusted_setup.g2_powers_of_tau
        ok = C.c_int(0)
        dummy = np.zeros(24, np.uint64)
        L.check(_decl().zk_kzg_verify(L.p64(np.ascontiguousarray(commitment, np.uint64)), L.p64(o) if o.size else L.p64(dummy), o.shape[0],
                                      L.p64(np.ascontiguousarray(proof.evaluation, np.uint64)), L.p64(prs) if prs.size else L.p64(dummy),
                                      prs.shape[0], L.p64(g2), g2.shape[0], C.byref(ok)))
        return bool(ok.value)
