"""Host mirror of circuit/src/arithmetic_circuit.rs and gkr/src/gkr_protocol.rs over the C ABI."""
import ctypes as C

import numpy as np

from . import _lib as L
from .mle import MultilinearPolynomial, limbs
from .sumcheck import SumcheckProverProof


class _Gate(C.Structure):
    _fields_ = [("left", C.c_uint64), ("right", C.c_uint64), ("out", C.c_uint64), ("op", C.c_uint64)]


class Operator:                                  # arithmetic_circuit.rs:5-8
    Add = 0
    Mul = 1


class Gate:                                      # :9-15
    def __init__(self, left_index, right_index, output_index, operator):
        self.left_index, self.right_index, self.output_index, self.operator = left_index, right_index, output_index, operator

    new = classmethod(lambda cls, l, r, o, op: cls(l, r, o, op))


class Layer:                                     # :17-19
    def __init__(self, gates):
        self.gates = list(gates)

    new = classmethod(lambda cls, gates: cls(gates))


def _decl():
    lib = L.lib()
    if getattr(lib, "_gkr_declared", False):
        return lib
    vp, sz, u64p = L.vp, L.sz, L.u64p
    gp, szp = C.POINTER(_Gate), C.POINTER(C.c_size_t)
    for name in ("zk_num_of_layer_variables", "zk_wiring_index", "zk_gkr_rounds", "zk_circuit_eval_size"):
        getattr(lib, name).restype = sz
    lib.zk_num_of_layer_variables.argtypes = [sz]
    lib.zk_gkr_rounds.argtypes = [sz]
    lib.zk_wiring_index.argtypes = [sz, sz, sz, sz]
    lib.zk_circuit_eval_size.argtypes = [gp, szp, sz, sz]
    sigs = {
        "zk_circuit_evaluate": [C.c_int, gp, szp, sz, u64p, sz, szp, u64p],
        "zk_circuit_add_mul_mle": [C.c_int, gp, sz, sz, C.POINTER(vp), C.POINTER(vp)],
        "zk_gkr_prove": [C.c_int, gp, szp, sz, u64p, sz, u64p, szp, u64p, u64p, u64p, u64p, u64p, u64p],
        "zk_gkr_verify": [C.c_int, gp, szp, sz, u64p, sz, u64p, sz, u64p, u64p, u64p, u64p, C.POINTER(C.c_int)],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib._gkr_declared = True
    return lib


def num_of_layer_variables(layer_index):         # :166
    return _decl().zk_num_of_layer_variables(layer_index)


def convert_to_binary_and_to_decimal(layer_index, variable_a, variable_b, variable_c):   # :180
    return _decl().zk_wiring_index(layer_index, variable_a, variable_b, variable_c)


class CircuitEvaluationResult:                   # :26-29
    def __init__(self, output, layer_evaluations):
        self.output, self.layer_evaluations = output, layer_evaluations


class Circuit:                                   # :21-24
    def __init__(self, field, layers):
        self.field = field
        self.layers = list(layers)

    new = classmethod(lambda cls, field, layers: cls(field, layers))

    def _flat(self, layers=None):
        layers = self.layers if layers is None else layers
        flat = [g for layer in layers for g in layer.gates]
        arr = (_Gate * max(len(flat), 1))(*[_Gate(g.left_index, g.right_index, g.output_index, g.operator) for g in flat])
        counts = (C.c_size_t * max(len(layers), 1))(*[len(layer.gates) for layer in layers])
        return arr, counts

    def evaluate(self, values):                  # :65-109
        lib = _decl()
        gates, counts = self._flat()
        x = np.ascontiguousarray(values, np.uint64).reshape(-1, limbs(self.field))
        nl = len(self.layers)
        tot = lib.zk_circuit_eval_size(gates, counts, nl, x.shape[0])
        sizes = (C.c_size_t * (nl + 1))()
        ev = np.zeros((tot, x.shape[1]), np.uint64)
        L.check(lib.zk_circuit_evaluate(self.field, gates, counts, nl, L.p64(x), x.shape[0], sizes, L.p64(ev)))
        out, off = [], 0
        for s in sizes:
            out.append(ev[off:off + s].copy())
            off += s
        return CircuitEvaluationResult(out[0], out)

    @staticmethod
    def w_i_polynomial(field, circuit_evaluation, layer_index):   # :114-124
        if layer_index >= len(circuit_evaluation.layer_evaluations):
            raise L.ReferencePanic(L.ZK_E_RANGE, "layer index out of bounds")
        return MultilinearPolynomial(field, circuit_evaluation.layer_evaluations[layer_index])

    def add_i_and_mul_i_mle(self, layer_index):  # :126-163
        gates, _ = self._flat([self.layers[layer_index]])
        a, m = C.c_void_p(), C.c_void_p()
        L.check(_decl().zk_circuit_add_mul_mle(self.field, gates, len(self.layers[layer_index].gates), layer_index,
                                               C.byref(a), C.byref(m)))
        return MultilinearPolynomial(self.field, _handle=a), MultilinearPolynomial(self.field, _handle=m)


class Proof:                                     # gkr_protocol.rs:17-23
    def __init__(self, circuit_output, claimed_sum, sumcheck_proofs, wb_evaluations, wc_evaluations, _flat=None):
        self.circuit_output = circuit_output
        self.claimed_sum = claimed_sum
        self.sumcheck_proofs = sumcheck_proofs
        self.wb_evaluations = wb_evaluations
        self.wc_evaluations = wc_evaluations
        self._flat = _flat


def prove(circuit, inputs):
    """gkr_protocol::prove :26-143"""
    lib = _decl()
    f = circuit.field
    Lm = limbs(f)
    gates, counts = circuit._flat()
    x = np.ascontiguousarray(inputs, np.uint64).reshape(-1, Lm)
    nl = len(circuit.layers)
    rounds = [lib.zk_gkr_rounds(i) for i in range(nl)]
    tot = sum(rounds)
    max_out = max([g.output_index for g in circuit.layers[0].gates] + [0]) + 1
    out = np.zeros((max_out, Lm), np.uint64)
    olen = C.c_size_t()
    cs = np.zeros(Lm, np.uint64)
    claims = np.zeros((nl, Lm), np.uint64)
    co = np.zeros((tot, 3, Lm), np.uint64)
    ch = np.zeros((tot, Lm), np.uint64)
    wb = np.zeros((max(nl - 1, 1), Lm), np.uint64)
    wc = np.zeros((max(nl - 1, 1), Lm), np.uint64)
    L.check(lib.zk_gkr_prove(f, gates, counts, nl, L.p64(x), x.shape[0], L.p64(out), C.byref(olen), L.p64(cs),
                             L.p64(claims), L.p64(co), L.p64(ch), L.p64(wb), L.p64(wc)))
    proofs, off = [], 0
    for i, r in enumerate(rounds):
        proofs.append(SumcheckProverProof(claims[i].copy(), co[off:off + r].copy(), ch[off:off + r].copy()))
        off += r
    return Proof(out[: olen.value], cs, proofs, wb[: nl - 1], wc[: nl - 1], _flat=(claims, co, ch))


def verify(circuit, proof, inputs):
    """gkr_protocol::verify :146-236"""
    lib = _decl()
    f = circuit.field
    Lm = limbs(f)
    gates, counts = circuit._flat()
    x = np.ascontiguousarray(inputs, np.uint64).reshape(-1, Lm)
    nl = len(circuit.layers)
    claims = np.ascontiguousarray(np.stack([p.claimed_sum for p in proof.sumcheck_proofs]), np.uint64)
    co = np.ascontiguousarray(np.concatenate([p.round_univariate_polynomials for p in proof.sumcheck_proofs]), np.uint64)
    pad = lambda a: np.ascontiguousarray(a, np.uint64) if len(a) else np.zeros((1, Lm), np.uint64)
    wb, wc = pad(proof.wb_evaluations), pad(proof.wc_evaluations)
    outp = np.ascontiguousarray(proof.circuit_output, np.uint64)
    ok = C.c_int(0)
    L.check(lib.zk_gkr_verify(f, gates, counts, nl, L.p64(x), x.shape[0], L.p64(outp), outp.shape[0], L.p64(claims),
                              L.p64(co), L.p64(wb), L.p64(wc), C.byref(ok)))
    return bool(ok.value)


class SuccinctProof(Proof):                      # succinct_gkr_protocol.rs:22-32
    def __init__(self, base, input_polynomial_commitment, input_rb_proof, input_rc_proof):
        super().__init__(base.circuit_output, base.claimed_sum, base.sumcheck_proofs, base.wb_evaluations,
                         base.wc_evaluations, _flat=base._flat)
        self.input_polynomial_commitment = input_polynomial_commitment
        self.input_rb_proof = input_rb_proof
        self.input_rc_proof = input_rc_proof


def prove_succinct(circuit, inputs, trusted_setup):
    """succinct_gkr_protocol::prove_succinct :35-169 (BLS12-381 Fr)"""
    from .kzg import MultilinearKZGProof, _decl as kzg_decl
    lib = _decl()
    kzg_decl()
    if not getattr(lib, "_succ_declared", False):
        u64p, sz, gp, szp = L.u64p, L.sz, C.POINTER(_Gate), C.POINTER(C.c_size_t)
        lib.zk_gkr_prove_succinct.argtypes = [gp, szp, sz, u64p, sz, L.vp, sz, u64p, szp] + [u64p] * 11
        lib.zk_gkr_prove_succinct.restype = C.c_int
        lib._succ_declared = True
    f = circuit.field
    if f != L.FR381:
        raise L.ZkError(L.ZK_E_ARG, "prove_succinct needs the pairing's scalar field (BLS12-381 Fr)")
    gates, counts = circuit._flat()
    x = np.ascontiguousarray(inputs, np.uint64).reshape(-1, 4)
    nl = len(circuit.layers)
    rounds = [lib.zk_gkr_rounds(i) for i in range(nl)]
    tot = sum(rounds)
    max_out = max([g.output_index for g in circuit.layers[0].gates] + [0]) + 1
    out = np.zeros((max_out, 4), np.uint64)
    olen = C.c_size_t()
    cs = np.zeros(4, np.uint64)
    claims = np.zeros((nl, 4), np.uint64)
    co = np.zeros((tot, 3, 4), np.uint64)
    ch = np.zeros((tot, 4), np.uint64)
    wb = np.zeros((max(nl - 1, 1), 4), np.uint64)
    wc = np.zeros((max(nl - 1, 1), 4), np.uint64)
    commit = np.zeros(12, np.uint64)
    rb_ev, rc_ev = np.zeros(4, np.uint64), np.zeros(4, np.uint64)
    rb_pr, rc_pr = np.zeros((nl, 12), np.uint64), np.zeros((nl, 12), np.uint64)
    L.check(lib.zk_gkr_prove_succinct(gates, counts, nl, L.p64(x), x.shape[0], trusted_setup.g1_powers_of_tau._h,
                                      trusted_setup.n_g2_powers_of_tau, L.p64(out), C.byref(olen), L.p64(cs), L.p64(claims),
                                      L.p64(co), L.p64(ch), L.p64(wb), L.p64(wc), L.p64(commit), L.p64(rb_ev), L.p64(rb_pr),
                                      L.p64(rc_ev), L.p64(rc_pr)))
    proofs, off = [], 0
    for i, r in enumerate(rounds):
        proofs.append(SumcheckProverProof(claims[i].copy(), co[off:off + r].copy(), ch[off:off + r].copy()))
        off += r
    base = Proof(out[: olen.value], cs, proofs, wb[: nl - 1], wc[: nl - 1], _flat=(claims, co, ch))
    return SuccinctProof(base, commit, MultilinearKZGProof(rb_ev, rb_pr), MultilinearKZGProof(rc_ev, rc_pr))


def verify_succinct(circuit, proof, trusted_setup):
    """succinct_gkr_protocol::verify_succinct :172-285"""
    lib = _decl()
    if not getattr(lib, "_succ_verify_declared", False):
        u64p, sz, gp, szp = L.u64p, L.sz, C.POINTER(_Gate), C.POINTER(C.c_size_t)
        lib.zk_gkr_verify_succinct.argtypes = [gp, szp, sz, u64p, sz, u64p, u64p, u64p, u64p, u64p, u64p, u64p, sz, u64p, u64p, sz, u64p, sz,
                                               C.POINTER(C.c_int)]
        lib.zk_gkr_verify_succinct.restype = C.c_int
        lib._succ_verify_declared = True
    if circuit.field != L.FR381:
        raise L.ZkError(L.ZK_E_ARG, "verify_succinct needs the pairing's scalar field (BLS12-381 Fr)")
    gates, counts = circuit._flat()
    nl = len(circuit.layers)
    claims = np.ascontiguousarray(np.stack([p.claimed_sum for p in proof.sumcheck_proofs]), np.uint64)
    co = np.ascontiguousarray(np.concatenate([p.round_univariate_polynomials for p in proof.sumcheck_proofs]), np.uint64)
    pad = lambda a, w: np.ascontiguousarray(a, np.uint64) if len(a) else np.zeros((1, w), np.uint64)
    wb, wc = pad(proof.wb_evaluations, 4), pad(proof.wc_evaluations, 4)
    outp = np.ascontiguousarray(proof.circuit_output, np.uint64)
    rb, rc = proof.input_rb_proof, proof.input_rc_proof
    rbp, rcp = pad(rb.proofs, 12), pad(rc.proofs, 12)
    g2 = trusted_setup.g2_powers_of_tau
    ok = C.c_int(0)
    L.check(lib.zk_gkr_verify_succinct(gates, counts, nl, L.p64(outp), outp.shape[0], L.p64(claims), L.p64(co), L.p64(wb), L.p64(wc),
                                       L.p64(np.ascontiguousarray(proof.input_polynomial_commitment, np.uint64)),
                                       L.p64(np.ascontiguousarray(rb.evaluation, np.uint64)), L.p64(rbp), len(rb.proofs),
                                       L.p64(np.ascontiguousarray(rc.evaluation, np.uint64)), L.p64(rcp), len(rc.proofs),
                                       L.p64(g2), g2.shape[0], C.byref(ok)))
    return bool(ok.value)


# ---- sparse (linear-time) GKR: layers are gate lists with their own widths (include/zkmle.h) -----------------
def _decl_sparse():
    lib = _decl()
    if getattr(lib, "_sparse_declared", False):
        return lib
    u64p, sz, gp, szp = L.u64p, L.sz, C.POINTER(_Gate), C.POINTER(C.c_size_t)
    u32p, fp = C.POINTER(C.c_uint32), C.POINTER(C.c_float)
    lib.zk_gkr_sparse_prove.argtypes = [C.c_int, gp, szp, sz, u32p, u64p, sz] + [u64p] * 8 + [fp]
    lib.zk_gkr_sparse_prove.restype = C.c_int
    lib.zk_gkr_sparse_wiring_eval.argtypes = [C.c_int, gp, sz, C.c_uint32, C.c_uint32] + [u64p] * 8
    lib.zk_gkr_sparse_wiring_eval.restype = C.c_int
    lib.zk_sparse_circuit_evaluate.argtypes = [C.c_int, gp, szp, sz, u32p, u64p, sz, u64p]
    lib.zk_sparse_circuit_evaluate.restype = C.c_int
    lib.zk_sparse_circuit_new.argtypes = [gp, szp, sz, u32p, sz, C.POINTER(L.vp)]
    lib.zk_sparse_circuit_new.restype = C.c_int
    lib.zk_sparse_circuit_free.argtypes = [L.vp]
    lib.zk_sparse_circuit_free.restype = C.c_int
    lib.zk_gkr_sparse_prove_compiled.argtypes = [C.c_int, L.vp, u64p, sz] + [u64p] * 8 + [fp]
    lib.zk_gkr_sparse_prove_compiled.restype = C.c_int
    lib._sparse_declared = True
    return lib


def gates_array(gate_rows):
    """(n, 4) uint64 array of (left, right, out, op) rows -> ctypes gate array (zero-copy for big circuits)"""
    rows = np.ascontiguousarray(gate_rows, np.uint64).reshape(-1, 4)
    return rows, rows.ctypes.data_as(C.POINTER(_Gate))


class SparseProof:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class SparseCircuit:
    """a circuit compiled once for the sparse prover (gate lists in HBM, grouped by left / right / output index)"""

    def __init__(self, layer_gate_rows, out_bits, ninputs):
        lib = _decl_sparse()
        nl = len(layer_gate_rows)
        rows = np.ascontiguousarray(np.concatenate([np.asarray(r, np.uint64).reshape(-1, 4) for r in layer_gate_rows]))
        counts = (C.c_size_t * nl)(*[len(r) for r in layer_gate_rows])
        ob = (C.c_uint32 * nl)(*out_bits)
        h = C.c_void_p()
        L.check(lib.zk_sparse_circuit_new(rows.ctypes.data_as(C.POINTER(_Gate)), counts, nl, ob, ninputs, C.byref(h)))
        self._h, self.out_bits, self.ninputs, self.nlayers = h, list(out_bits), ninputs, nl

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                L.lib().zk_sparse_circuit_free(self._h)
            except Exception:
                pass
            self._h = None


def sparse_prove(field, layer_gate_rows, out_bits, inputs, circuit=None):
    """layer_gate_rows: list (layer 0 = output layer) of (n_l, 4) arrays of (left, right, out, op);
    pass a SparseCircuit as `circuit` to reuse a compiled circuit."""
    lib = _decl_sparse()
    Lm = limbs(field)
    x = np.ascontiguousarray(inputs, np.uint64).reshape(-1, Lm)
    if circuit is None:
        circuit = SparseCircuit(layer_gate_rows, out_bits, x.shape[0])
    out_bits = circuit.out_bits
    nl = circuit.nlayers
    in_bits = [out_bits[i + 1] if i + 1 < nl else x.shape[0].bit_length() - 1 for i in range(nl)]
    rounds = [2 * k for k in in_bits]
    tot = sum(rounds)
    out = np.zeros((1 << out_bits[0], Lm), np.uint64)
    cs = np.zeros(Lm, np.uint64)
    claims = np.zeros((nl, Lm), np.uint64)
    co = np.zeros((tot, 3, Lm), np.uint64)
    ch = np.zeros((tot, Lm), np.uint64)
    wb = np.zeros((max(nl - 1, 1), Lm), np.uint64)
    wc = np.zeros((max(nl - 1, 1), Lm), np.uint64)
    ra = np.zeros((out_bits[0], Lm), np.uint64)
    ms = (C.c_float * nl)()
    L.check(lib.zk_gkr_sparse_prove_compiled(field, circuit._h, L.p64(x), x.shape[0], L.p64(out), L.p64(cs), L.p64(claims), L.p64(co),
                                             L.p64(ch), L.p64(wb), L.p64(wc), L.p64(ra), ms))
    return SparseProof(circuit_output=out, claimed_sum=cs, layer_claims=claims, coeffs=co, challenges=ch, wb_evals=wb[: nl - 1],
                       wc_evals=wc[: nl - 1], output_challenges=ra, rounds=rounds, in_bits=in_bits, ms_layers=list(ms))


def sparse_wiring_eval(field, gate_rows, out_bits, in_bits, pa, rb, rc, alpha=None, beta=None, pb=None):
    lib = _decl_sparse()
    Lm = limbs(field)
    rows = np.ascontiguousarray(gate_rows, np.uint64).reshape(-1, 4)
    a, m = np.zeros(Lm, np.uint64), np.zeros(Lm, np.uint64)
    c = lambda v: L.p64(np.ascontiguousarray(v, np.uint64)) if v is not None else None
    L.check(lib.zk_gkr_sparse_wiring_eval(field, rows.ctypes.data_as(C.POINTER(_Gate)), rows.shape[0], out_bits, in_bits, c(alpha), c(pa),
                                          c(beta), c(pb), c(rb), c(rc), L.p64(a), L.p64(m)))
    return a, m


def _fe_to_bytes_be(field, a):
    out = np.zeros(8 * limbs(field), np.uint8)
    L.check(L.lib().zk_fe_to_bytes_be(field, L.p64(np.ascontiguousarray(a, np.uint64)), L.p8(out)))
    return out.tobytes()


def sparse_verify(field, layer_gate_rows, out_bits, proof, inputs):
    """The verifier of gkr_protocol.rs:146-236 for the sparse representation (the wiring predicates are
    evaluated from the gate lists in O(#gates) on the GPU).  Used by tests and the config-4 bench."""
    from .mle import MultilinearPolynomial
    from .sumcheck import Transcript, SumcheckProverProof, verify as sumcheck_verify
    from . import sharded as S
    S._declare_host()
    lib = L.lib()
    Lm = limbs(field)

    def mul(a, b):
        o = np.zeros(Lm, np.uint64)
        L.check(lib.zk_fe_mul(field, L.p64(np.ascontiguousarray(a)), L.p64(np.ascontiguousarray(b)), L.p64(o)))
        return o

    nl = len(layer_gate_rows)
    t = Transcript()
    w0 = MultilinearPolynomial(field, proof.circuit_output)
    t.append(w0.convert_to_bytes())
    ra = np.stack([t.random_challenge_as_field_element(field) for _ in range(out_bits[0])])
    claim = w0.evaluate(ra)
    alpha = beta = None
    pa, pb = ra, None
    off = 0
    x = MultilinearPolynomial(field, inputs)
    for l in range(nl):
        if not np.array_equal(claim, proof.layer_claims[l]):
            return False
        r = proof.rounds[l]
        res = sumcheck_verify(SumcheckProverProof(proof.layer_claims[l], proof.coeffs[off:off + r], None), t, field)
        if not res.is_proof_valid:
            return False
        ch = res.random_challenges
        k = r // 2
        if l + 1 < nl:
            wb, wc = proof.wb_evals[l], proof.wc_evals[l]
        else:
            wb, wc = x.evaluate(ch[:k]), x.evaluate(ch[k:])
        add_r, mul_r = sparse_wiring_eval(field, layer_gate_rows[l], out_bits[l], k, pa, ch[:k], ch[k:], alpha, beta, pb)
        expect = S.fe_add(field, mul(add_r, S.fe_add(field, wb, wc)), mul(mul_r, mul(wb, wc)))
        if not np.array_equal(expect, res.last_claimed_sum):
            return False
        t.append(_fe_to_bytes_be(field, wb))
        alpha = t.random_challenge_as_field_element(field)
        t.append(_fe_to_bytes_be(field, wc))
        beta = t.random_challenge_as_field_element(field)
        claim = S.fe_add(field, mul(alpha, wb), mul(beta, wc))
        pa, pb = ch[:k], ch[k:]
        off += r
    return True
