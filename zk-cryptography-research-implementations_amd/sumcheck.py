"""Host mirror of the reference's transcript, composed polynomials and sumcheck provers over the C ABI.

  Transcript                      transcripts/src/fiat_shamir/fiat_shamir_transcript.rs
  ProductPolynomial, SumPolynomial polynomials/src/composed/{product,sum}_polynomial.rs
  Prover / Verifier / SumcheckProof sumcheck_protocol/src/basic_sumcheck/{prover,verifier}.rs
  prove / verify / generate_round_univariate  sumcheck_protocol/src/gkr_sumcheck/sumcheck_gkr_protocol.rs
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .mle import MultilinearPolynomial, limbs


class SumcheckStats(C.Structure):
    _fields_ = [("rounds", C.c_uint32), ("ms_absorb", C.c_float), ("ms_rounds", C.c_float)]


def _decl():
    lib = L.lib()
    if getattr(lib, "_sumcheck_declared", False):
        return lib
    vp, sz, u64p, u8p = L.vp, L.sz, L.u64p, L.u8p
    sigs = {
        "zk_sumcheck_last_stats": [C.POINTER(SumcheckStats)],
        "zk_transcript_new": [C.POINTER(vp)], "zk_transcript_free": [vp],
        "zk_transcript_append": [vp, u8p, sz], "zk_transcript_sample": [vp, u8p],
        "zk_transcript_challenge": [vp, C.c_int, u64p], "zk_keccak256": [u8p, sz, u8p],
        "zk_transcript_export_state": [vp, u64p, C.POINTER(C.c_uint32)], "zk_transcript_import_state": [vp, u64p, C.c_uint32],
        "zk_uni_evaluate": [C.c_int, u64p, sz, u64p, u64p],
        "zk_uni_lagrange_interpolate": [C.c_int, u64p, u64p, sz, u64p],
        "zk_sumcheck_basic_prove": [vp, u64p, u64p, u64p],
        "zk_sumcheck_basic_prove_on": [vp, vp, u64p, u64p, u64p],
        "zk_sumcheck_basic_verify": [vp, u64p, u64p, sz, C.POINTER(C.c_int)],
        "zk_sumpoly_evaluate": [C.POINTER(vp), sz, sz, u64p, sz, u64p],
        "zk_sumpoly_reduce": [C.POINTER(vp), sz, sz, vp],
        "zk_prodpoly_reduce": [C.POINTER(vp), sz, vp],
        "zk_sumpoly_round_evals": [C.POINTER(vp), sz, sz, u64p],
        "zk_sumcheck_gkr_prove": [C.POINTER(vp), sz, sz, u64p, vp, u64p, u64p],
        "zk_sumcheck_gkr_rounds": [C.POINTER(vp), sz, sz, vp, u64p, u64p, u64p],
        "zk_sumcheck_gkr_rounds_cf": [C.POINTER(vp), sz, sz, u64p, vp, u64p, u64p, u64p],
        "zk_sumcheck_gkr_verify": [C.c_int, u64p, u64p, sz, sz, vp, u64p, u64p, C.POINTER(C.c_int)],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib._sumcheck_declared = True
    return lib


def last_stats():
    """Host-clock split of the last prover call on this thread: {'rounds', 'ms_absorb', 'ms_rounds'}."""
    st = SumcheckStats()
    L.check(_decl().zk_sumcheck_last_stats(C.byref(st)))
    return {"rounds": st.rounds, "ms_absorb": st.ms_absorb, "ms_rounds": st.ms_rounds}


def _bytes_arr(data):
    data = bytes(data)
    return (np.frombuffer(data, np.uint8).copy() if data else np.zeros(1, np.uint8)), len(data)


def keccak256(data):
    buf, n = _bytes_arr(data)
    out = np.zeros(32, np.uint8)
    L.check(_decl().zk_keccak256(L.p8(buf), n, L.p8(out)))
    return out.tobytes()


class Transcript:
    def __init__(self):
        h = C.c_void_p()
        L.check(_decl().zk_transcript_new(C.byref(h)))
        self._h = h

    @classmethod
    def new(cls):
        return cls()

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                L.lib().zk_transcript_free(self._h)
            except Exception:
                pass
            self._h = None

    def append(self, incoming_data):
        buf, n = _bytes_arr(incoming_data)
        L.check(_decl().zk_transcript_append(self._h, L.p8(buf), n))

    def sample_random_challenge(self):
        out = np.zeros(32, np.uint8)
        L.check(_decl().zk_transcript_sample(self._h, L.p8(out)))
        return out.tobytes()

    def random_challenge_as_field_element(self, field):
        out = np.zeros(limbs(field), np.uint64)
        L.check(_decl().zk_transcript_challenge(self._h, field, L.p64(out)))
        return out

    def export_state(self):
        """the running sponge: 26 uint64 words = 25 Keccak lanes + the fill of the open block (208 bytes)"""
        st = np.zeros(26, np.uint64)
        fill = C.c_uint32(0)
        L.check(_decl().zk_transcript_export_state(self._h, L.p64(st), C.byref(fill)))
        st[25] = fill.value
        return st

    def import_state(self, st):
        st = np.ascontiguousarray(st, np.uint64)
        L.check(_decl().zk_transcript_import_state(self._h, L.p64(st), int(st[25])))


def lagrange_interpolate(field, xs, ys):
    xs = np.ascontiguousarray(xs, np.uint64).reshape(-1, limbs(field))
    ys = np.ascontiguousarray(ys, np.uint64).reshape(-1, limbs(field))
    out = np.zeros_like(xs)
    L.check(_decl().zk_uni_lagrange_interpolate(field, L.p64(xs), L.p64(ys), xs.shape[0], L.p64(out)))
    return out


def uni_evaluate(field, coeffs, x):
    c = np.ascontiguousarray(coeffs, np.uint64).reshape(-1, limbs(field))
    out = np.zeros(limbs(field), np.uint64)
    L.check(_decl().zk_uni_evaluate(field, L.p64(c), c.shape[0], L.p64(np.ascontiguousarray(x, np.uint64)), L.p64(out)))
    return out


# ---- composed polynomials -----------------------------------------------------------------------
class ProductPolynomial:
    def __init__(self, polynomials):
        polynomials = list(polynomials)
        n0 = polynomials[0].number_of_variables()
        if any(p.number_of_variables() != n0 for p in polynomials):     # product_polynomial.rs:16-21
            raise L.ReferencePanic(L.ZK_E_NVARS, "different number of variables")
        self.polynomials = polynomials

    new = classmethod(lambda cls, polynomials: cls(polynomials))

    def degree(self):                                                   # :85-87
        return len(self.polynomials)

    def partial_evaluate(self, evaluating_variable, value):             # :36-54
        return [MultilinearPolynomial.partial_evaluate(p, evaluating_variable, value) for p in self.polynomials]

    def evaluate(self, values):                                         # :26-34
        return SumPolynomial._evaluate([self], values)

    def multiply_polynomials_element_wise(self):                        # :58-73
        if len(self.polynomials) < 2:
            raise L.ReferencePanic(L.ZK_E_NEED_TWO, "more than one polynomial required for mul operation")
        out = MultilinearPolynomial.alloc(self.polynomials[0].field, len(self.polynomials[0]))
        arr = (C.c_void_p * len(self.polynomials))(*[p._h for p in self.polynomials])
        L.check(_decl().zk_prodpoly_reduce(arr, len(self.polynomials), out._h))
        return out


class SumPolynomial:
    def __init__(self, product_polynomials):
        product_polynomials = list(product_polynomials)
        n0 = product_polynomials[0].polynomials[0].number_of_variables()
        if any(p.number_of_variables() != n0 for pp in product_polynomials for p in pp.polynomials):   # sum_polynomial.rs:17-23
            raise L.ReferencePanic(L.ZK_E_NVARS, "different number of variables")
        self.product_polynomials = product_polynomials

    new = classmethod(lambda cls, product_polynomials: cls(product_polynomials))

    def degree(self):                                                   # :88-90
        return self.product_polynomials[0].degree()

    def number_of_variables(self):                                      # :92-94
        return self.product_polynomials[0].polynomials[0].number_of_variables()

    @property
    def field(self):
        return self.product_polynomials[0].polynomials[0].field

    def _handles(self):
        nfac = self.degree()
        flat = []
        for pp in self.product_polynomials:
            if len(pp.polynomials) != nfac:
                raise L.ZkError(L.ZK_E_ARG, "every product must have the same number of factors")
            flat += [p._h for p in pp.polynomials]
        return (C.c_void_p * len(flat))(*flat), len(self.product_polynomials), nfac

    @staticmethod
    def _evaluate(products, values):
        sp = SumPolynomial.__new__(SumPolynomial)
        sp.product_polynomials = products
        arr, nprod, nfac = sp._handles()
        f = sp.field
        vals = np.ascontiguousarray(values, np.uint64).reshape(-1, limbs(f))
        out = np.zeros(limbs(f), np.uint64)
        L.check(_decl().zk_sumpoly_evaluate(arr, nprod, nfac, L.p64(vals), vals.shape[0], L.p64(out)))
        return out

    @staticmethod
    def _reduce(list_of_lists):
        flat = [p._h for lst in list_of_lists for p in lst]
        arr = (C.c_void_p * len(flat))(*flat)
        first = list_of_lists[0][0]
        out = MultilinearPolynomial.alloc(first.field, len(first))
        L.check(_decl().zk_sumpoly_reduce(arr, len(list_of_lists), len(list_of_lists[0]), out._h))
        return out

    def evaluate(self, values):                                         # :30-38
        return SumPolynomial._evaluate(self.product_polynomials, values)

    def partial_evaluate(self, evaluating_variable, value):             # :40-53
        return SumPolynomial([ProductPolynomial(pp.partial_evaluate(evaluating_variable, value))
                              for pp in self.product_polynomials])

    def add_polynomials_element_wise(self):                             # :57-76
        if len(self.product_polynomials) < 2:
            raise L.ReferencePanic(L.ZK_E_NEED_TWO, "more than one product polynomial required for add operation")
        arr, nprod, nfac = self._handles()
        first = self.product_polynomials[0].polynomials[0]
        out = MultilinearPolynomial.alloc(first.field, len(first))
        L.check(_decl().zk_sumpoly_reduce(arr, nprod, nfac, out._h))
        return out


def generate_round_univariate(current_polynomial):
    """sumcheck_gkr_protocol.rs:113-143 : evaluations at 0..=degree, one fused pass over the tables"""
    arr, nprod, nfac = current_polynomial._handles()
    f = current_polynomial.field
    out = np.zeros((nfac + 1, limbs(f)), np.uint64)
    L.check(_decl().zk_sumpoly_round_evals(arr, nprod, nfac, L.p64(out)))
    return out


class SumcheckProverProof:                                              # sumcheck_gkr_protocol.rs:8-13
    def __init__(self, claimed_sum, round_univariate_polynomials, random_challenges):
        self.claimed_sum = claimed_sum
        self.round_univariate_polynomials = round_univariate_polynomials   # (rounds, degree+1, limbs) coefficients
        self.random_challenges = random_challenges


class SumcheckVerifierProof:                                            # :15-20
    def __init__(self, is_proof_valid, random_challenges, last_claimed_sum):
        self.is_proof_valid = is_proof_valid
        self.random_challenges = random_challenges
        self.last_claimed_sum = last_claimed_sum


def prove(sum_polynomial, claimed_sum, transcript):
    """sumcheck_gkr_protocol::prove :24-67"""
    arr, nprod, nfac = sum_polynomial._handles()
    f = sum_polynomial.field
    n = sum_polynomial.number_of_variables()
    co = np.zeros((max(n, 1), nfac + 1, limbs(f)), np.uint64)
    ch = np.zeros((max(n, 1), limbs(f)), np.uint64)
    cs = np.ascontiguousarray(claimed_sum, np.uint64).reshape(-1)
    L.check(_decl().zk_sumcheck_gkr_prove(arr, nprod, nfac, L.p64(cs), transcript._h, L.p64(co), L.p64(ch)))
    return SumcheckProverProof(cs.copy(), co[:n], ch[:n])


def gkr_rounds_const_factors(field, tables, const_factors, transcript):
    """the rounds of `prove` (:37-60, no claimed-sum append) on two-factor products whose second factor may be a constant:
    tables[p] = (MultilinearPolynomial, MultilinearPolynomial or None), const_factors[p] used where the second one is None.
    -> (coefficient rows, challenges, final values (2 per product))"""
    nprod = len(tables)
    flat = [t for prod in tables for t in prod]
    arr = (C.c_void_p * len(flat))(*[t._h if t is not None else None for t in flat])
    n = len(flat[0]).bit_length() - 1
    Lm = limbs(field)
    co = np.zeros((max(n, 1), 3, Lm), np.uint64)
    ch = np.zeros((max(n, 1), Lm), np.uint64)
    fin = np.zeros((2 * nprod, Lm), np.uint64)
    cf = np.ascontiguousarray(const_factors, np.uint64).reshape(nprod, Lm)
    L.check(_decl().zk_sumcheck_gkr_rounds_cf(arr, nprod, 2, L.p64(cf), transcript._h, L.p64(co), L.p64(ch), L.p64(fin)))
    return co[:n], ch[:n], fin


def verify(proof, transcript, field):
    """sumcheck_gkr_protocol::verify :69-105"""
    co = np.ascontiguousarray(proof.round_univariate_polynomials, np.uint64)
    nr, nc = co.shape[0], co.shape[1]
    ch = np.zeros((max(nr, 1), limbs(field)), np.uint64)
    last = np.zeros(limbs(field), np.uint64)
    ok = C.c_int(0)
    cs = np.ascontiguousarray(proof.claimed_sum, np.uint64).reshape(-1)
    L.check(_decl().zk_sumcheck_gkr_verify(field, L.p64(cs), L.p64(co) if co.size else L.p64(ch), nr, nc,
                                           transcript._h, L.p64(ch), L.p64(last), C.byref(ok)))
    if not ok.value:
        return SumcheckVerifierProof(False, np.zeros((0, limbs(field)), np.uint64), last)   # :85-89
    return SumcheckVerifierProof(True, ch[:nr], last)


# ---- basic sumcheck -------------------------------------------------------------------------------
class SumcheckProof:                                                    # prover.rs:15-19
    def __init__(self, initial_polynomial, initial_claimed_sum, round_univariate_polynomials):
        self.initial_polynomial = initial_polynomial
        self.initial_claimed_sum = initial_claimed_sum
        self.round_univariate_polynomials = round_univariate_polynomials   # (nvars, 2, limbs)


class Prover:
    """basic_sumcheck::prover::Prover (prover.rs:7-71); the table lives in HBM."""

    def __init__(self):
        self.is_initialized = False

    @classmethod
    def init(cls, field, polynomial_evaluated_values):                  # :22-33
        self = cls()
        if isinstance(polynomial_evaluated_values, MultilinearPolynomial):
            self.initial_polynomial = polynomial_evaluated_values
        else:
            self.initial_polynomial = MultilinearPolynomial(field, polynomial_evaluated_values)
        self.field = field
        self.initial_claimed_sum = self.initial_polynomial.sum()        # :28
        self.transcript = Transcript()                                  # :24, a public field of the reference's Prover (:10)
        self.round_univariate_polynomials = None
        self.challenges = None
        self.is_initialized = True
        return self

    def prove(self):                                                    # :35-71
        if not self.is_initialized:
            raise L.ReferencePanic(L.ZK_E_NOT_INIT, "Can't prove without init")
        n = self.initial_polynomial.number_of_variables()
        Lm = limbs(self.field)
        cs = np.zeros(Lm, np.uint64)
        rp = np.zeros((max(n, 1), 2, Lm), np.uint64)
        ch = np.zeros((max(n, 1), Lm), np.uint64)
        L.check(_decl().zk_sumcheck_basic_prove_on(self.initial_polynomial._h, self.transcript._h, L.p64(cs), L.p64(rp), L.p64(ch)))   # appends to self.transcript (:38-58)
        assert np.array_equal(cs, self.initial_claimed_sum)
        self.round_univariate_polynomials = rp[:n]
        self.challenges = ch[:n]
        return SumcheckProof(self.initial_polynomial, cs, rp[:n])


class Verifier:
    """basic_sumcheck::verifier::Verifier (verifier.rs:8-71)"""

    def __init__(self):
        self.is_initialized = False

    @classmethod
    def init(cls):
        self = cls()
        self.is_initialized = True
        return self

    def verify(self, proof):
        if not self.is_initialized:
            raise L.ReferencePanic(L.ZK_E_NOT_INIT, "Can't verify without init")
        rp = np.ascontiguousarray(proof.round_univariate_polynomials, np.uint64)
        nr = rp.shape[0] if rp.size else 0
        ok = C.c_int(0)
        cs = np.ascontiguousarray(proof.initial_claimed_sum, np.uint64)
        ptr = L.p64(rp) if rp.size else L.p64(cs)
        L.check(_decl().zk_sumcheck_basic_verify(proof.initial_polynomial._h, L.p64(cs), ptr, nr, C.byref(ok)))
        return bool(ok.value)
