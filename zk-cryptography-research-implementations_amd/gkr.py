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
