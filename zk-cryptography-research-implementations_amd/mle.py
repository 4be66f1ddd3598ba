"""MultilinearPolynomial: host mirror of polynomials/src/multilinear/evaluation_form.rs over HBM tables."""
import ctypes as C

import numpy as np

from . import _lib as L


def limbs(field):
    n = L.lib().zk_field_limbs(field)
    if n < 0:
        raise L.ZkError(n, "bad field")
    return n


def from_ints(field, values):
    """canonical Python ints (any sign) -> Montgomery numpy array (len, limbs)"""
    n = limbs(field)
    p = MODULI[field]
    canon = np.zeros((len(values), n), np.uint64)
    for i, v in enumerate(values):
        v = int(v) % p
        for k in range(n):
            canon[i, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    out = np.zeros_like(canon)
    if len(values):
        L.check(L.lib().zk_vec_from_canonical(field, L.p64(canon), len(values), L.p64(out)))
    return out


def to_ints(field, arr):
    n = limbs(field)
    arr = np.ascontiguousarray(arr, np.uint64).reshape(-1, n)
    canon = np.zeros_like(arr)
    if arr.shape[0]:
        L.check(L.lib().zk_vec_to_canonical(field, L.p64(arr), arr.shape[0], L.p64(canon)))
    return [sum(int(x) << (64 * k) for k, x in enumerate(row)) for row in canon]


MODULI = {
    L.FR381: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    L.FQ381: 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
    L.BN254_FQ: 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,
    L.BN254_FR: 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001,
}


def _elem(field, value):
    a = np.ascontiguousarray(value, np.uint64).reshape(-1)
    if a.shape[0] != limbs(field):
        raise L.ZkError(L.ZK_E_ARG, "element has the wrong number of limbs")
    return a


class MultilinearPolynomial:
    """`MultilinearPolynomial<F>{evaluated_values}` (evaluation_form.rs:7-9) resident in HBM.

    Index bit (n-1-v) <-> variable v; elements are arkworks-layout Montgomery limbs.
    """

    def __init__(self, field, evaluated_values=None, _handle=None):
        self.field = field
        if _handle is not None:
            self._h = _handle
            return
        vals = np.ascontiguousarray(evaluated_values, np.uint64).reshape(-1, limbs(field))
        h = C.c_void_p()
        # new(): asserts a power-of-two length (:13) and copies the slice (:16) -- here the copy is the upload
        if vals.shape[0] == 0:
            raise L.ReferencePanic(L.ZK_E_NOT_POW2, "Evaluated values must be a power of 2")
        L.check(L.lib().zk_table_upload(field, L.p64(vals), vals.shape[0], C.byref(h)))
        self._h = h

    # -- construction helpers ------------------------------------------------------------------
    @classmethod
    def new(cls, field, evaluated_values):
        return cls(field, evaluated_values)

    @classmethod
    def from_ints(cls, field, ints):
        return cls(field, from_ints(field, ints))

    @classmethod
    def vector(cls, field, values):
        """a plain HBM vector of any length >= 1 (no power-of-two assert): MSM scalars"""
        vals = np.ascontiguousarray(values, np.uint64).reshape(-1, limbs(field))
        h = C.c_void_p()
        L.check(L.lib().zk_table_upload_raw(field, L.p64(vals), vals.shape[0], C.byref(h)))
        return cls(field, _handle=h)

    @classmethod
    def alloc(cls, field, length):
        h = C.c_void_p()
        L.check(L.lib().zk_table_alloc(field, length, C.byref(h)))
        return cls(field, _handle=h)

    @classmethod
    def random(cls, field, length, seed):
        """synthetic table generated on the device (SURVEY 8d generator)"""
        t = cls.alloc(field, length)
        L.check(L.lib().zk_table_fill_random(t._h, seed))
        return t

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                L.lib().zk_table_free(h)
            except Exception:
                pass
            self._h = None

    def __len__(self):
        return L.lib().zk_table_len(self._h)

    @property
    def device_ptr(self):
        return L.lib().zk_table_device_ptr(self._h)

    @property
    def evaluated_values(self):
        out = np.zeros((len(self), limbs(self.field)), np.uint64)
        L.check(L.lib().zk_table_download(self._h, L.p64(out)))
        return out

    def to_ints(self):
        return to_ints(self.field, self.evaluated_values)

    def clone(self):
        h = C.c_void_p()
        L.check(L.lib().zk_table_clone(self._h, C.byref(h)))
        return MultilinearPolynomial(self.field, _handle=h)

    # -- the reference's methods ---------------------------------------------------------------
    def number_of_variables(self):                          # :45
        return len(self).bit_length() - 1

    @staticmethod
    def partial_evaluate(polynomial, evaluating_variable, value, stream=None):   # :61
        n = len(polynomial)
        if n < 2:
            raise L.ReferencePanic(L.ZK_E_NOT_POW2, "Evaluated values must be a power of 2")
        out = MultilinearPolynomial.alloc(polynomial.field, n // 2)
        L.check(L.lib().zk_mle_fold(polynomial._h, evaluating_variable, L.p64(_elem(polynomial.field, value)),
                                    out._h, stream))
        return out

    def evaluate(self, values):                             # :21
        vals = np.ascontiguousarray(values, np.uint64).reshape(-1, limbs(self.field))
        out = np.zeros(limbs(self.field), np.uint64)
        ptr = L.p64(vals) if vals.size else L.p64(np.zeros(limbs(self.field), np.uint64))
        L.check(L.lib().zk_mle_evaluate(self._h, ptr, vals.shape[0], L.p64(out)))
        return out

    def convert_to_bytes(self):                             # :35
        out = np.zeros(len(self) * 8 * limbs(self.field), np.uint8)
        L.check(L.lib().zk_mle_to_bytes(self._h, L.p8(out)))
        return out.tobytes()

    def scalar_mul(self, scalar, stream=None):              # :49
        out = MultilinearPolynomial.alloc(self.field, len(self))
        L.check(L.lib().zk_mle_scalar_mul(self._h, L.p64(_elem(self.field, scalar)), out._h, stream))
        return out

    def sub_scalar(self, scalar, stream=None):              # multilinear_kzg.rs:74-78
        out = MultilinearPolynomial.alloc(self.field, len(self))
        L.check(L.lib().zk_mle_sub_scalar(self._h, L.p64(_elem(self.field, scalar)), out._h, stream))
        return out

    @staticmethod
    def add_polynomials(poly1, poly2, stream=None):         # :145
        out = MultilinearPolynomial.alloc(poly1.field, max(len(poly1), 1))
        L.check(L.lib().zk_mle_add(poly1._h, poly2._h, out._h, stream))
        return out

    @staticmethod
    def polynomial_tensor_add(w_b, w_c, stream=None):       # :108
        out = MultilinearPolynomial.alloc(w_b.field, len(w_b) * len(w_c))
        L.check(L.lib().zk_mle_tensor_add(w_b._h, w_c._h, out._h, stream))
        return out

    @staticmethod
    def polynomial_tensor_mul(w_b, w_c, stream=None):       # :125
        out = MultilinearPolynomial.alloc(w_b.field, len(w_b) * len(w_c))
        L.check(L.lib().zk_mle_tensor_mul(w_b._h, w_c._h, out._h, stream))
        return out

    # -- sums (prover.rs:28 and :74-89) --------------------------------------------------------
    def sum(self):
        out = np.zeros(limbs(self.field), np.uint64)
        L.check(L.lib().zk_mle_sum(self._h, L.p64(out)))
        return out

    def half_sums(self):
        out = np.zeros((2, limbs(self.field)), np.uint64)
        L.check(L.lib().zk_mle_half_sums(self._h, L.p64(out)))
        return out

    def fold_half_sums(self, value, stream=None):
        """fused sumcheck round: (partial_evaluate(self, 0, value), its two half sums)"""
        out = MultilinearPolynomial.alloc(self.field, len(self) // 2)
        sums = np.zeros((2, limbs(self.field)), np.uint64)
        L.check(L.lib().zk_mle_fold_half_sums(self._h, L.p64(_elem(self.field, value)), out._h, L.p64(sums), stream))
        return out, sums
