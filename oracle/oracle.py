"""ctypes binding of the CPU oracle (libzkoracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package never does.  Elements travel as numpy uint64 arrays of shape (..., limbs)
in Montgomery form (the arkworks in-memory layout); helpers convert to / from Python ints.
"""
import ctypes as C
import os
import subprocess

import numpy as np

FR381, FQ381, BN254_FQ, BN254_FR = 0, 1, 2, 3
FIELD_NAMES = {FR381: "bls12_381_fr", FQ381: "bls12_381_fq", BN254_FQ: "bn254_fq", BN254_FR: "bn254_fr"}
E_NOT_POW2, E_LEN_MISMATCH, E_NVARS, E_NEED_TWO, E_KZG_LEN, E_RANGE, E_ARG, E_NOMEM = range(-1, -9, -1)

_HERE = os.path.dirname(os.path.abspath(__file__))
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)
_szp = C.POINTER(C.c_size_t)


class OraclePanic(Exception):
    """The reference would panic here (status code in .code)."""

    def __init__(self, code, where=""):
        super().__init__(f"oracle status {code} {where}")
        self.code = code


def build(force=False):
    lib = os.path.join(_HERE, "libzkoracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("field.c", "mle.c", "gkr.c", "g1.c", "zkoracle.h", "zk_internal.h")]
    if force or not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "libzkoracle.so"], stdout=subprocess.DEVNULL)
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("ZKORACLE_LIB") or build()
        _lib = C.CDLL(path)
        _lib.orc_transcript_new.restype = C.c_void_p
        _lib.orc_bench_fold.restype = C.c_double
        _lib.orc_bench_commit_naive.restype = C.c_double
        _lib.orc_bench_pippenger_mt.restype = C.c_double
        _lib.orc_bench_fold_mt.restype = C.c_double
        for n in ("orc_num_layer_variables", "orc_wiring_index", "orc_circuit_eval_size", "orc_gkr_rounds"):
            getattr(_lib, n).restype = C.c_size_t
    return _lib


def _p(a):
    return a.ctypes.data_as(_u64p)


def _b(a):
    return a.ctypes.data_as(_u8p)


def _chk(rc, where=""):
    if rc < 0:
        raise OraclePanic(rc, where)
    return rc


def limbs(field):
    return lib().orc_field_limbs(field)


def constants(field):
    n = limbs(field)
    m, r, r2 = (np.zeros(n, np.uint64) for _ in range(3))
    inv = C.c_uint64()
    _chk(lib().orc_field_constants(field, _p(m), _p(r), _p(r2), C.byref(inv)))
    return limbs_to_int(m), limbs_to_int(r), limbs_to_int(r2), inv.value


def limbs_to_int(a):
    return sum(int(x) << (64 * i) for i, x in enumerate(np.asarray(a).reshape(-1)))


def int_to_limbs(v, n):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def modulus(field):
    return constants(field)[0]


def from_ints(field, values):
    """canonical Python ints (any sign / size) -> Montgomery array (len, limbs)"""
    n = limbs(field)
    p = modulus(field)
    canon = np.zeros((len(values), n), np.uint64)
    for i, v in enumerate(values):
        canon[i] = int_to_limbs(int(v) % p, n)
    out = np.zeros_like(canon)
    _chk(lib().orc_vec_from_canonical(field, _p(canon), len(values), _p(out)))
    return out


def to_ints(field, arr):
    n = limbs(field)
    arr = np.ascontiguousarray(arr, np.uint64).reshape(-1, n)
    canon = np.zeros_like(arr)
    _chk(lib().orc_vec_to_canonical(field, _p(arr), arr.shape[0], _p(canon)))
    return [limbs_to_int(row) for row in canon]


def _arr(field, a):
    a = np.ascontiguousarray(a, np.uint64)
    return a.reshape(-1, limbs(field))


def fe_op(field, name, a, b=None):
    out = np.zeros(limbs(field), np.uint64)
    fn = getattr(lib(), "orc_fe_" + name)
    a = _arr(field, a)
    if b is None:
        _chk(fn(field, _p(a), _p(out)))
    else:
        b = _arr(field, b)
        _chk(fn(field, _p(a), _p(b), _p(out)))
    return out


def from_le_bytes_mod_order(field, data):
    out = np.zeros(limbs(field), np.uint64)
    buf = np.frombuffer(bytes(data), np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
    _chk(lib().orc_fe_from_le_bytes_mod_order(field, _b(buf), len(data), _p(out)))
    return out


def keccak256(data):
    out = np.zeros(32, np.uint8)
    buf = np.frombuffer(bytes(data), np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
    lib().orc_keccak256(_b(buf), C.c_size_t(len(data)), _b(out))
    return out.tobytes()


class Transcript:
    """transcripts/src/fiat_shamir/fiat_shamir_transcript.rs"""

    def __init__(self):
        self.h = C.c_void_p(lib().orc_transcript_new())

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_transcript_free(self.h)
            self.h = None

    def append(self, data):
        buf = np.frombuffer(bytes(data), np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
        lib().orc_transcript_append(self.h, _b(buf), C.c_size_t(len(data)))

    def sample_random_challenge(self):
        out = np.zeros(32, np.uint8)
        lib().orc_transcript_sample(self.h, _b(out))
        return out.tobytes()

    def random_challenge_as_field_element(self, field):
        out = np.zeros(limbs(field), np.uint64)
        _chk(lib().orc_transcript_challenge(self.h, field, _p(out)))
        return out


# ---- MultilinearPolynomial -------------------------------------------------------------------
def mle_new_check(length):
    _chk(lib().orc_mle_new_check(C.c_size_t(length)), "MultilinearPolynomial::new")


def partial_evaluate(field, poly, var, value):
    poly = _arr(field, poly)
    out = np.zeros((max(poly.shape[0] // 2, 1), poly.shape[1]), np.uint64)
    _chk(lib().orc_mle_partial_evaluate(field, _p(poly), C.c_size_t(poly.shape[0]), C.c_size_t(var),
                                        _p(_arr(field, value)), _p(out)), "partial_evaluate")
    return out[: poly.shape[0] // 2]


def evaluate(field, poly, values):
    poly = _arr(field, poly)
    values = _arr(field, values) if len(values) else np.zeros((0, limbs(field)), np.uint64)
    out = np.zeros(limbs(field), np.uint64)
    vp = _p(values) if values.size else _p(np.zeros(limbs(field), np.uint64))
    _chk(lib().orc_mle_evaluate(field, _p(poly), C.c_size_t(poly.shape[0]), vp,
                                C.c_size_t(values.shape[0]), _p(out)), "evaluate")
    return out


def mle_to_bytes(field, poly):
    poly = _arr(field, poly)
    out = np.zeros(poly.shape[0] * 8 * poly.shape[1], np.uint8)
    _chk(lib().orc_mle_to_bytes(field, _p(poly), C.c_size_t(poly.shape[0]), _b(out)))
    return out.tobytes()


def fe_to_bytes_be(field, a):
    out = np.zeros(8 * limbs(field), np.uint8)
    _chk(lib().orc_fe_to_bytes_be(field, _p(_arr(field, a)), _b(out)))
    return out.tobytes()


def fe_to_bytes_le(field, a):
    out = np.zeros(8 * limbs(field), np.uint8)
    _chk(lib().orc_fe_to_bytes_le(field, _p(_arr(field, a)), _b(out)))
    return out.tobytes()


def scalar_mul(field, poly, s):
    poly = _arr(field, poly)
    out = np.zeros_like(poly)
    _chk(lib().orc_mle_scalar_mul(field, _p(poly), C.c_size_t(poly.shape[0]), _p(_arr(field, s)), _p(out)))
    return out


def add_polynomials(field, a, b):
    a, b = _arr(field, a), _arr(field, b)
    out = np.zeros_like(a)
    _chk(lib().orc_mle_add(field, _p(a), C.c_size_t(a.shape[0]), _p(b), C.c_size_t(b.shape[0]), _p(out)),
         "add_polynomials")
    return out


def _tensor(field, wb, wc, name):
    wb, wc = _arr(field, wb), _arr(field, wc)
    out = np.zeros((wb.shape[0] * wc.shape[0], wb.shape[1]), np.uint64)
    _chk(getattr(lib(), name)(field, _p(wb), C.c_size_t(wb.shape[0]), _p(wc), C.c_size_t(wc.shape[0]), _p(out)),
         name)
    return out


def polynomial_tensor_add(field, wb, wc):
    return _tensor(field, wb, wc, "orc_mle_tensor_add")


def polynomial_tensor_mul(field, wb, wc):
    return _tensor(field, wb, wc, "orc_mle_tensor_mul")


def vec_sum(field, a):
    a = _arr(field, a)
    out = np.zeros(limbs(field), np.uint64)
    _chk(lib().orc_vec_sum(field, _p(a), C.c_size_t(a.shape[0]), _p(out)))
    return out


# ---- univariate ------------------------------------------------------------------------------
def uni_evaluate(field, coeffs, x):
    c = _arr(field, coeffs)
    out = np.zeros(limbs(field), np.uint64)
    _chk(lib().orc_uni_evaluate(field, _p(c), C.c_size_t(c.shape[0]), _p(_arr(field, x)), _p(out)))
    return out


def lagrange_interpolate(field, xs, ys):
    xs, ys = _arr(field, xs), _arr(field, ys)
    out = np.zeros_like(xs)
    _chk(lib().orc_uni_lagrange_interpolate(field, _p(xs), _p(ys), C.c_size_t(xs.shape[0]), _p(out)))
    return out


# ---- basic sumcheck --------------------------------------------------------------------------
def split_and_sum(field, table):
    t = _arr(field, table)
    out = np.zeros((2, t.shape[1]), np.uint64)
    _chk(lib().orc_split_and_sum(field, _p(t), C.c_size_t(t.shape[0]), _p(out)))
    return out


def sumcheck_basic_prove(field, table):
    """-> (claimed_sum, round_polys (n,2,limbs), challenges (n,limbs))"""
    t = _arr(field, table)
    n = max(int(t.shape[0]).bit_length() - 1, 0)
    L = t.shape[1]
    claimed = np.zeros(L, np.uint64)
    rounds = np.zeros((max(n, 1), 2, L), np.uint64)
    chal = np.zeros((max(n, 1), L), np.uint64)
    _chk(lib().orc_sumcheck_basic_prove(field, _p(t), C.c_size_t(t.shape[0]), _p(claimed), _p(rounds), _p(chal)),
         "Prover::prove")
    return claimed, rounds[:n], chal[:n]


def sumcheck_basic_verify(field, table, claimed_sum, round_polys):
    t = _arr(field, table)
    rp = np.ascontiguousarray(round_polys, np.uint64)
    nr = rp.reshape(-1, 2, t.shape[1]).shape[0] if rp.size else 0
    rc = lib().orc_sumcheck_basic_verify(field, _p(t), C.c_size_t(t.shape[0]), _p(_arr(field, claimed_sum)),
                                         _p(rp) if rp.size else _p(np.zeros(1, np.uint64)), C.c_size_t(nr))
    return bool(_chk(rc, "Verifier::verify"))


# ---- composed / GKR sumcheck ------------------------------------------------------------------
def _tables(field, tables):
    """tables: array (nprod, nfac, len, limbs)"""
    t = np.ascontiguousarray(tables, np.uint64)
    assert t.ndim == 4
    return t


def sumpoly_evaluate(field, tables, values):
    t = _tables(field, tables)
    v = _arr(field, values)
    out = np.zeros(t.shape[3], np.uint64)
    _chk(lib().orc_sumpoly_evaluate(field, _p(t), C.c_size_t(t.shape[0]), C.c_size_t(t.shape[1]),
                                    C.c_size_t(t.shape[2]), _p(v), C.c_size_t(v.shape[0]), _p(out)))
    return out


def sumpoly_reduce(field, tables):
    t = _tables(field, tables)
    out = np.zeros((t.shape[2], t.shape[3]), np.uint64)
    _chk(lib().orc_sumpoly_reduce(field, _p(t), C.c_size_t(t.shape[0]), C.c_size_t(t.shape[1]),
                                  C.c_size_t(t.shape[2]), _p(out)), "add_polynomials_element_wise")
    return out


def gkr_round_univariate(field, tables):
    t = _tables(field, tables)
    out = np.zeros((t.shape[1] + 1, t.shape[3]), np.uint64)
    _chk(lib().orc_gkr_round_univariate(field, _p(t), C.c_size_t(t.shape[0]), C.c_size_t(t.shape[1]),
                                        C.c_size_t(t.shape[2]), _p(out)), "generate_round_univariate")
    return out


def sumcheck_gkr_prove(field, tables, claimed_sum, transcript):
    """-> (round_coeffs (n, nfac+1, limbs), challenges (n, limbs))"""
    t = _tables(field, tables)
    n = int(t.shape[2]).bit_length() - 1
    L = t.shape[3]
    co = np.zeros((max(n, 1), t.shape[1] + 1, L), np.uint64)
    ch = np.zeros((max(n, 1), L), np.uint64)
    _chk(lib().orc_sumcheck_gkr_prove(field, _p(t), C.c_size_t(t.shape[0]), C.c_size_t(t.shape[1]),
                                      C.c_size_t(t.shape[2]), _p(_arr(field, claimed_sum)), transcript.h,
                                      _p(co), _p(ch)), "sumcheck_gkr::prove")
    return co[:n], ch[:n]


def sumcheck_gkr_verify(field, claimed_sum, round_coeffs, transcript):
    """-> (is_valid, challenges, last_claimed_sum)"""
    co = np.ascontiguousarray(round_coeffs, np.uint64)
    nr, nc, L = co.shape
    ch = np.zeros((max(nr, 1), L), np.uint64)
    last = np.zeros(L, np.uint64)
    ok = _chk(lib().orc_sumcheck_gkr_verify(field, _p(_arr(field, claimed_sum)), _p(co), C.c_size_t(nr),
                                            C.c_size_t(nc), transcript.h, _p(ch), _p(last)))
    return bool(ok), ch[:nr], last


# ---- circuit / GKR ----------------------------------------------------------------------------
class Gate(C.Structure):
    _fields_ = [("left", C.c_uint64), ("right", C.c_uint64), ("out", C.c_uint64), ("op", C.c_uint64)]


ADD, MUL = 0, 1


def _circuit(layers):
    """layers: list (layer 0 = output layer) of lists of (left, right, out, op)"""
    flat = [g for layer in layers for g in layer]
    arr = (Gate * max(len(flat), 1))(*[Gate(*g) for g in flat])
    counts = (C.c_size_t * max(len(layers), 1))(*[len(layer) for layer in layers])
    return arr, counts


def num_of_layer_variables(layer_index):
    return lib().orc_num_layer_variables(C.c_size_t(layer_index))


def wiring_index(layer_index, a, b, c):
    return lib().orc_wiring_index(C.c_size_t(layer_index), C.c_size_t(a), C.c_size_t(b), C.c_size_t(c))


def circuit_evaluate(field, layers, inputs):
    """-> list of layer evaluations, index 0 = output, last = inputs"""
    gates, counts = _circuit(layers)
    x = _arr(field, inputs)
    tot = lib().orc_circuit_eval_size(gates, counts, C.c_size_t(len(layers)), C.c_size_t(x.shape[0]))
    sizes = (C.c_size_t * (len(layers) + 1))()
    ev = np.zeros((tot, x.shape[1]), np.uint64)
    _chk(lib().orc_circuit_evaluate(field, gates, counts, C.c_size_t(len(layers)), _p(x), C.c_size_t(x.shape[0]),
                                    sizes, _p(ev)), "Circuit::evaluate")
    out, off = [], 0
    for s in sizes:
        out.append(ev[off:off + s].copy())
        off += s
    return out


def add_i_and_mul_i_mle(field, layer_gates, layer_index):
    gates, _ = _circuit([layer_gates])
    n = 1 << num_of_layer_variables(layer_index)
    L = limbs(field)
    a, m = np.zeros((n, L), np.uint64), np.zeros((n, L), np.uint64)
    _chk(lib().orc_circuit_add_mul_mle(field, gates, C.c_size_t(len(layer_gates)), C.c_size_t(layer_index),
                                       _p(a), _p(m)), "add_i_and_mul_i_mle")
    return a, m


def gkr_rounds(layer_index):
    return lib().orc_gkr_rounds(C.c_size_t(layer_index))


def gkr_prove(field, layers, inputs):
    gates, counts = _circuit(layers)
    x = _arr(field, inputs)
    L = x.shape[1]
    nl = len(layers)
    tot_rounds = sum(gkr_rounds(i) for i in range(nl))
    max_out = max([g[2] for g in layers[0]] + [0]) + 1
    proof = dict(
        circuit_output=np.zeros((max_out, L), np.uint64), claimed_sum=np.zeros(L, np.uint64),
        layer_claims=np.zeros((nl, L), np.uint64), coeffs=np.zeros((tot_rounds, 3, L), np.uint64),
        challenges=np.zeros((tot_rounds, L), np.uint64), wb_evals=np.zeros((max(nl - 1, 1), L), np.uint64),
        wc_evals=np.zeros((max(nl - 1, 1), L), np.uint64))
    olen = C.c_size_t()
    _chk(lib().orc_gkr_prove(field, gates, counts, C.c_size_t(nl), _p(x), C.c_size_t(x.shape[0]),
                             _p(proof["circuit_output"]), C.byref(olen), _p(proof["claimed_sum"]),
                             _p(proof["layer_claims"]), _p(proof["coeffs"]), _p(proof["challenges"]),
                             _p(proof["wb_evals"]), _p(proof["wc_evals"])), "gkr::prove")
    proof["circuit_output"] = proof["circuit_output"][: olen.value]
    proof["wb_evals"] = proof["wb_evals"][: nl - 1]
    proof["wc_evals"] = proof["wc_evals"][: nl - 1]
    return proof


def gkr_prove_wide(field, rows, out_bits, inputs):
    """gkr_protocol::prove for layers of any width, from the definition, linear in the gates (gkr_wide.c).  rows[l]: uint64 array
    (ngates, 4) of (left, right, out, op) -- the layout the product's sparse prover takes; out_bits[l] = log2 of layer l's width."""
    x = _arr(field, inputs)
    L = x.shape[1]
    nl = len(rows)
    flat = np.ascontiguousarray(np.concatenate([np.asarray(r, np.uint64).reshape(-1, 4) for r in rows]))
    counts = (C.c_size_t * nl)(*[len(r) for r in rows])
    ob = (C.c_uint32 * nl)(*[int(b) for b in out_bits])
    widths = [1 << int(b) for b in out_bits] + [x.shape[0]]
    rounds = [2 * (widths[l + 1].bit_length() - 1) for l in range(nl)]
    tot = sum(rounds)
    proof = dict(
        circuit_output=np.zeros((widths[0], L), np.uint64), claimed_sum=np.zeros(L, np.uint64),
        layer_claims=np.zeros((nl, L), np.uint64), coeffs=np.zeros((max(tot, 1), 3, L), np.uint64),
        challenges=np.zeros((max(tot, 1), L), np.uint64), wb_evals=np.zeros((max(nl - 1, 1), L), np.uint64),
        wc_evals=np.zeros((max(nl - 1, 1), L), np.uint64), output_challenges=np.zeros((int(out_bits[0]), L), np.uint64))
    _chk(lib().orc_gkr_prove_wide(field, flat.ctypes.data_as(C.c_void_p), counts, C.c_size_t(nl), ob, _p(x), C.c_size_t(x.shape[0]),
                                  _p(proof["circuit_output"]), _p(proof["claimed_sum"]), _p(proof["layer_claims"]), _p(proof["coeffs"]),
                                  _p(proof["challenges"]), _p(proof["wb_evals"]), _p(proof["wc_evals"]), _p(proof["output_challenges"])),
         "gkr::prove (wide)")
    proof["coeffs"] = proof["coeffs"][:tot]
    proof["challenges"] = proof["challenges"][:tot]
    proof["wb_evals"] = proof["wb_evals"][: nl - 1]
    proof["wc_evals"] = proof["wc_evals"][: nl - 1]
    return proof


def gkr_verify(field, layers, proof, inputs):
    gates, counts = _circuit(layers)
    x = _arr(field, inputs)
    pad = lambda a: a if a.size else np.zeros((1, x.shape[1]), np.uint64)
    rc = lib().orc_gkr_verify(field, gates, counts, C.c_size_t(len(layers)), _p(x), C.c_size_t(x.shape[0]),
                              _p(proof["circuit_output"]), C.c_size_t(proof["circuit_output"].shape[0]),
                              _p(proof["layer_claims"]), _p(proof["coeffs"]), _p(proof["challenges"]),
                              _p(pad(proof["wb_evals"])), _p(pad(proof["wc_evals"])))
    return bool(_chk(rc, "gkr::verify"))


# ---- G1 / KZG ---------------------------------------------------------------------------------
def g1_generator():
    out = np.zeros(12, np.uint64)
    lib().orc_g1_generator(_p(out))
    return out


def g1_is_on_curve(p):
    return bool(lib().orc_g1_is_on_curve(_p(np.ascontiguousarray(p, np.uint64))))


def g1_add(p, q):
    out = np.zeros(12, np.uint64)
    lib().orc_g1_add(_p(np.ascontiguousarray(p, np.uint64)), _p(np.ascontiguousarray(q, np.uint64)), _p(out))
    return out


def g1_neg(p):
    out = np.zeros(12, np.uint64)
    lib().orc_g1_neg(_p(np.ascontiguousarray(p, np.uint64)), _p(out))
    return out


def g1_mul_fr(p, scalar):
    out = np.zeros(12, np.uint64)
    lib().orc_g1_mul_fr(_p(np.ascontiguousarray(p, np.uint64)), _p(np.ascontiguousarray(scalar, np.uint64)), _p(out))
    return out


def g1_affine_ints(p):
    """affine point (12 limbs, Montgomery) -> (x, y) canonical ints, or None for infinity"""
    p = np.ascontiguousarray(p, np.uint64).reshape(2, 6)
    if not p.any():
        return None
    x, y = to_ints(FQ381, p)
    return x, y


def kzg_lagrange_basis(taus):
    t = _arr(FR381, taus)
    out = np.zeros((1 << t.shape[0], 4), np.uint64)
    _chk(lib().orc_kzg_lagrange_basis(_p(t), C.c_size_t(t.shape[0]), _p(out)))
    return out


def kzg_setup_g1(taus):
    t = _arr(FR381, taus)
    out = np.zeros((1 << t.shape[0], 12), np.uint64)
    _chk(lib().orc_kzg_setup_g1(_p(t), C.c_size_t(t.shape[0]), _p(out)))
    return out


def kzg_commit(values, points):
    v = _arr(FR381, values)
    pts = np.ascontiguousarray(points, np.uint64).reshape(-1, 12)
    out = np.zeros(12, np.uint64)
    _chk(lib().orc_kzg_commit(_p(v), C.c_size_t(v.shape[0]), _p(pts), C.c_size_t(pts.shape[0]), _p(out)),
         "commit_to_polynomial")
    return out


def kzg_open(values, points, opening, n_g2=None):
    v = _arr(FR381, values)
    pts = np.ascontiguousarray(points, np.uint64).reshape(-1, 12)
    o = _arr(FR381, opening)
    ev = np.zeros(4, np.uint64)
    proofs = np.zeros((max(o.shape[0], 1), 12), np.uint64)
    _chk(lib().orc_kzg_open(_p(v), C.c_size_t(v.shape[0]), _p(pts), C.c_size_t(pts.shape[0]), _p(o),
                            C.c_size_t(o.shape[0]), C.c_size_t(o.shape[0] if n_g2 is None else n_g2),
                            _p(ev), _p(proofs)), "open_and_prove")
    return ev, proofs[: o.shape[0]]


def kzg_quotients(values, opening):
    v = _arr(FR381, values)
    o = _arr(FR381, opening)
    out = np.zeros((max(v.shape[0] - 1, 1), 4), np.uint64)
    _chk(lib().orc_kzg_quotients(_p(v), C.c_size_t(v.shape[0]), _p(o), C.c_size_t(o.shape[0]), _p(out)))
    res, off, n = [], 0, v.shape[0]
    for _ in range(o.shape[0]):
        n //= 2
        res.append(out[off:off + n].copy())
        off += n
    return res


def gkr_prove_succinct(layers, inputs, g1_points, n_g2=None):
    """prove_succinct, gkr/src/succinct_gkr_protocol.rs:35-169: commit(inputs) :42-44, the layer loop of
    gkr_protocol::prove (rb / rc are taken from EVERY layer's challenges here, :121-126, so after the loop
    they are the last layer's), then open(inputs, rb) and open(inputs, rc) :154-157."""
    x = _arr(FR381, inputs)
    proof = gkr_prove(FR381, layers, x)
    proof["input_polynomial_commitment"] = kzg_commit(x, g1_points)
    last = gkr_rounds(len(layers) - 1)
    ch = proof["challenges"][-last:]
    mid = last // 2
    proof["input_rb_proof"] = kzg_open(x, g1_points, ch[:mid], n_g2)
    proof["input_rc_proof"] = kzg_open(x, g1_points, ch[mid:], n_g2)
    return proof


# ---- cpu baseline -----------------------------------------------------------------------------
def bench_fold(field, table, r, reps):
    t = _arr(field, table)
    return lib().orc_bench_fold(field, _p(t), C.c_size_t(t.shape[0]), _p(_arr(field, r)), reps)


def bench_fold_mt(field, table, r, reps):
    """-> (seconds, threads used): OpenMP over the output indices"""
    t = _arr(field, table)
    used = C.c_int(0)
    secs = lib().orc_bench_fold_mt(field, _p(t), C.c_size_t(t.shape[0]), _p(_arr(field, r)), reps, C.byref(used))
    return secs, used.value


def bench_commit_naive(values, points):
    v = _arr(FR381, values)
    pts = np.ascontiguousarray(points, np.uint64).reshape(-1, 12)
    return lib().orc_bench_commit_naive(_p(v), C.c_size_t(v.shape[0]), _p(pts))


def msm_pippenger(values, points, window_bits=8, slices=4):
    """best-effort CPU bucket method (all cores); same point as kzg_commit.  -> affine point (12 limbs)"""
    v = _arr(FR381, values)
    pts = np.ascontiguousarray(points, np.uint64).reshape(-1, 12)
    out = np.zeros(12, np.uint64)
    used = C.c_int(0)
    _chk(lib().orc_msm_pippenger(_p(v), C.c_size_t(v.shape[0]), _p(pts), window_bits, slices, _p(out), C.byref(used)), "msm_pippenger")
    return out


def bench_pippenger_mt(values, points, window_bits=None, slices=None):
    """-> (seconds, threads used, window bits): the CPU Pippenger over all host cores"""
    import os
    v = _arr(FR381, values)
    pts = np.ascontiguousarray(points, np.uint64).reshape(-1, 12)
    n = v.shape[0]
    cores = os.cpu_count() or 1
    if window_bits is None:
        window_bits = max(4, min(12, n.bit_length() - 8))
    if slices is None:
        w = (255 + window_bits - 1) // window_bits
        slices = max(1, (2 * cores + w - 1) // w)
    out = np.zeros(12, np.uint64)
    used = C.c_int(0)
    secs = lib().orc_bench_pippenger_mt(_p(v), C.c_size_t(n), _p(pts), window_bits, slices, _p(out), C.byref(used))
    if secs < 0:
        raise OraclePanic(E_ARG, "orc_bench_pippenger_mt")
    return secs, used.value, window_bits
