"""Oracle pin 2: every known-answer value the reference's own #[test]s hold for the path
(tests/golden/reference_kats.json, each entry cites file:line), run through BOTH restatements
(C oracle and the Python big-int model), plus the [derived] Appendix-B vectors."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import pymodel as M

FID = {"bn254_fq": O.BN254_FQ, "bls12_381_fr": O.FR381, "bls12_381_fq": O.FQ381, "bn254_fr": O.BN254_FR}
OPS = {"add": O.ADD, "mul": O.MUL}


def enc(field, vals):
    return O.from_ints(FID[field], vals)


def dec(field, arr):
    return O.to_ints(FID[field], arr)


def norm(field, vals):
    p = M.P[field]
    return [v % p for v in vals]


def layers_of(spec):
    return [[(g[0], g[1], g[2], OPS[g[3]]) for g in layer] for layer in spec]


def test_partial_evaluate(ref_kats):
    for k in ref_kats["partial_evaluate"]:
        f = k["field"]
        got = O.partial_evaluate(FID[f], enc(f, k["poly"]), k["var"], enc(f, [k["value"]]))
        assert dec(f, got) == k["expect"], k["src"]
        assert M.partial_evaluate(k["poly"], k["var"], k["value"], M.P[f]) == k["expect"]


def test_evaluate(ref_kats):
    for k in ref_kats["evaluate"]:
        f = k["field"]
        assert dec(f, O.evaluate(FID[f], enc(f, k["poly"]), enc(f, k["values"]))) == [k["expect"]]
        assert M.evaluate(k["poly"], k["values"], M.P[f]) == k["expect"]


def test_new_panics_on_non_power_of_two(ref_kats):
    for k in ref_kats["new_panics"]:
        with pytest.raises(O.OraclePanic) as e:
            O.mle_new_check(k["len"])
        assert e.value.code == O.E_NOT_POW2
    for n in (1, 2, 4, 1 << 20):
        O.mle_new_check(n)
    with pytest.raises(O.OraclePanic):
        O.mle_new_check(0)


def test_evaluate_with_too_many_values_panics():
    f = "bn254_fq"
    with pytest.raises(O.OraclePanic) as e:   # folding a 1-entry table -> pow-2 assert
        O.evaluate(FID[f], enc(f, [0, 0, 3, 8]), enc(f, [6, 2, 9]))
    assert e.value.code == O.E_NOT_POW2
    # fewer values than variables: returns element 0 of the partially folded table (no length check)
    assert dec(f, O.evaluate(FID[f], enc(f, [0, 0, 3, 8]), enc(f, [6]))) == [18]


def test_tensor_ops(ref_kats):
    for k in ref_kats["tensor_add"]:
        f = k["field"]
        assert dec(f, O.polynomial_tensor_add(FID[f], enc(f, k["wb"]), enc(f, k["wc"]))) == k["expect"]
    for k in ref_kats["tensor_mul"]:
        f = k["field"]
        assert dec(f, O.polynomial_tensor_mul(FID[f], enc(f, k["wb"]), enc(f, k["wc"]))) == k["expect"]
    for k in ref_kats["tensor_mul_panics"]:
        f = k["field"]
        with pytest.raises(O.OraclePanic) as e:
            O.polynomial_tensor_mul(FID[f], enc(f, k["wb"]), enc(f, k["wc"]))
        assert e.value.code == O.E_LEN_MISMATCH


def _tables(field, products):
    return np.stack([np.stack([enc(field, t) for t in prod]) for prod in products])


def test_product_and_sum_polynomial(ref_kats):
    k = ref_kats["product_poly"]
    f = k["field"]
    # a ProductPolynomial alone = SumPolynomial with one product, for evaluate
    t = _tables(f, [k["polys"]])
    assert dec(f, O.sumpoly_evaluate(FID[f], t, enc(f, k["evaluate"]["values"]))) == [k["evaluate"]["expect"]]
    pe = k["partial_evaluate"]
    for poly, exp in zip(k["polys"], pe["expect"]):
        assert dec(f, O.partial_evaluate(FID[f], enc(f, poly), pe["var"], enc(f, [pe["value"]]))) == exp
    s = ref_kats["sum_poly"]
    t = _tables(f, s["products"])
    assert dec(f, O.sumpoly_evaluate(FID[f], t, enc(f, s["evaluate"]["values"]))) == [s["evaluate"]["expect"]]
    assert dec(f, O.sumpoly_reduce(FID[f], t)) == s["add_element_wise"]
    # product [0,0,0,6] is the element-wise reduce of a (poly, poly) + zero product
    zero = [[0, 0, 0, 0], [0, 0, 0, 0]]
    assert dec(f, O.sumpoly_reduce(FID[f], _tables(f, [k["polys"], zero]))) == k["multiply_element_wise"]
    pe = s["partial_evaluate"]
    for prod, eprod in zip(s["products"], pe["expect"]):
        for poly, exp in zip(prod, eprod):
            assert dec(f, O.partial_evaluate(FID[f], enc(f, poly), pe["var"], enc(f, [pe["value"]]))) == exp
    with pytest.raises(O.OraclePanic) as e:   # ">1 polynomial required"
        O.sumpoly_reduce(FID[f], _tables(f, [k["polys"]]))
    assert e.value.code == O.E_NEED_TWO


def test_gkr_round_univariate(ref_kats):
    for k in ref_kats["gkr_round_univariate"]:
        f = k["field"]
        assert dec(f, O.gkr_round_univariate(FID[f], _tables(f, k["products"]))) == k["expect"]
        assert M.round_evals(k["products"], M.P[f]) == k["expect"]


def test_gkr_sumcheck_roundtrip_and_derived(ref_kats, derived_kats):
    for k in ref_kats["gkr_sumcheck_roundtrip"]:
        f = k["field"]
        tp, tv = O.Transcript(), O.Transcript()
        co, ch = O.sumcheck_gkr_prove(FID[f], _tables(f, k["products"]), enc(f, [k["claimed_sum"]]), tp)
        ok, vch, last = O.sumcheck_gkr_verify(FID[f], enc(f, [k["claimed_sum"]]), co, tv)
        assert ok == k["expect_valid"]
        assert np.array_equal(vch, ch)
        # the oracle check the caller performs: f(challenges) == last claimed sum
        assert dec(f, O.sumpoly_evaluate(FID[f], _tables(f, k["products"]), ch)) == dec(f, last)
    d = derived_kats["gkr_sumcheck"]
    f = d["field"]
    co, ch = O.sumcheck_gkr_prove(FID[f], _tables(f, d["products"]), enc(f, [d["claimed"]]), O.Transcript())
    assert [[hex(x) for x in dec(f, r)] for r in co] == [[hex(int(x, 16)) for x in r] for r in d["coeffs"]]
    assert [hex(x) for x in dec(f, ch)] == [hex(int(x, 16)) for x in d["challenges"]]
    mp, mc = M.sumcheck_gkr_prove(d["products"], d["claimed"], M.Transcript(), M.P[f])
    assert mp == [dec(f, r) for r in co] and mc == dec(f, ch)
    # a wrong claim is rejected
    ok, _, _ = O.sumcheck_gkr_verify(FID[f], enc(f, [13]), co, O.Transcript())
    assert not ok


def test_basic_sumcheck(ref_kats, derived_kats):
    for k in ref_kats["basic_sumcheck_claimed_sum"]:
        f = k["field"]
        cs, _, _ = O.sumcheck_basic_prove(FID[f], enc(f, k["table"]))
        assert dec(f, cs) == [k["expect"]]
    for k in ref_kats["basic_sumcheck_roundtrip"]:
        f = k["field"]
        table = k.get("table") or [k["table_constant"]] * (1 << min(k["log_len"], 12))
        t = enc(f, table)
        cs, rp, ch = O.sumcheck_basic_prove(FID[f], t)
        assert O.sumcheck_basic_verify(FID[f], t, cs, rp) is True
        mcs, mrp, mch = M.sumcheck_basic_prove(table, M.P[f])
        assert dec(f, cs) == [mcs] and [dec(f, r) for r in rp] == mrp and dec(f, ch) == mch
        assert M.sumcheck_basic_verify(table, mcs, mrp, M.P[f])
        # tampering is caught
        bad = rp.copy()
        bad[0, 0, 0] ^= np.uint64(1)
        assert O.sumcheck_basic_verify(FID[f], t, cs, bad) is False
    for d in derived_kats["basic_sumcheck"]:
        f = d["field"]
        cs, rp, ch = O.sumcheck_basic_prove(FID[f], enc(f, d["table"]))
        assert dec(f, cs) == [int(d["claimed"], 16)]
        for got, exp in zip(rp, d["rounds"]):
            assert dec(f, got) == [int(x, 16) for x in exp]
        assert dec(f, ch)[: len(d["challenges"])] == [int(x, 16) for x in d["challenges"]]
        assert dec(f, O.evaluate(FID[f], enc(f, d["table"]), ch)) == [int(d["final_eval"], 16)]


def test_univariate(ref_kats):
    for k in ref_kats["lagrange_interpolate"]:
        f = k["field"]
        assert dec(f, O.lagrange_interpolate(FID[f], enc(f, k["xs"]), enc(f, k["ys"]))) == k["expect"]
        assert M.lagrange_interpolate(k["xs"], k["ys"], M.P[f]) == k["expect"]
    for k in ref_kats["univariate_evaluate"]:
        f = k["field"]
        assert dec(f, O.uni_evaluate(FID[f], enc(f, k["coeffs"]), enc(f, [k["x"]]))) == [k["expect"]]


def test_circuit(ref_kats):
    for k in ref_kats["circuit_evaluate"]:
        f = k["field"]
        evs = O.circuit_evaluate(FID[f], layers_of(k["layers"]), enc(f, k["inputs"]))
        mevs = M.circuit_evaluate(layers_of(k["layers"]), k["inputs"], M.P[f])
        assert [dec(f, e) for e in evs] == mevs
        if "expect_layers" in k:
            assert mevs == k["expect_layers"], k["src"]
        else:
            assert mevs[0] == k["expect_output"]
    for li, nv in ref_kats["num_of_layer_variables"]["expect"].items():
        assert O.num_of_layer_variables(int(li)) == nv == M.num_vars(int(li))
    for k in ref_kats["add_i_mul_i"]:
        gates = [(g[0], g[1], g[2], OPS[g[3]]) for g in k["gates"]]
        a, m = O.add_i_and_mul_i_mle(O.BN254_FQ, gates, k["layer_index"])
        assert a.shape[0] == k["len"]
        av, mv = dec("bn254_fq", a), dec("bn254_fq", m)
        assert [i for i, v in enumerate(av) if v] == k["add_ones"] and set(av) <= {0, 1}
        assert [i for i, v in enumerate(mv) if v] == k["mul_ones"] and set(mv) <= {0, 1}
        ma, mm = M.add_mul_mle(gates, k["layer_index"])
        assert ma == av and mm == mv


def test_gkr_roundtrip_and_derived(ref_kats, derived_kats):
    for k in ref_kats["gkr_roundtrip"]:
        f = k["field"]
        L = layers_of(k["layers"])
        x = enc(f, k["inputs"])
        proof = O.gkr_prove(FID[f], L, x)
        assert O.gkr_verify(FID[f], L, proof, x) is True, k["src"]
        mp = M.gkr_prove(L, k["inputs"], M.P[f])
        assert M.gkr_verify(L, mp, k["inputs"], M.P[f])
        assert dec(f, proof["claimed_sum"]) == [mp["claimed_sum"]]
        assert dec(f, proof["wb_evals"]) == mp["wb"] and dec(f, proof["wc_evals"]) == mp["wc"]
        flat = [c for sp in mp["sumcheck_proofs"] for poly in sp["polys"] for c in poly]
        assert dec(f, proof["coeffs"]) == flat
        # wrong inputs are rejected by the verifier's own input evaluation
        bad = enc(f, [v + 1 for v in k["inputs"]])
        assert O.gkr_verify(FID[f], L, proof, bad) is False
    for d in derived_kats["gkr_prove"]:
        f = d["field"]
        proof = O.gkr_prove(FID[f], layers_of(d["layers"]), enc(f, d["inputs"]))
        assert dec(f, proof["circuit_output"]) == d["output"]
        assert dec(f, proof["claimed_sum"]) == [int(d["claimed_sum"], 16)]
        if "layer0_claim" in d:
            assert dec(f, proof["layer_claims"][0]) == [int(d["layer0_claim"], 16)]
            assert dec(f, proof["wb_evals"]) == [int(x, 16) for x in d["wb"]]
            assert dec(f, proof["wc_evals"]) == [int(x, 16) for x in d["wc"]]


def test_kzg_lagrange_basis(ref_kats):
    for k in ref_kats["kzg_lagrange_basis"]:
        got = O.to_ints(O.FR381, O.kzg_lagrange_basis(O.from_ints(O.FR381, k["taus"])))
        assert got == norm("bls12_381_fr", k["expect"]) == M.lagrange_basis(k["taus"])
