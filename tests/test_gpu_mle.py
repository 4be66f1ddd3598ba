"""GPU parity (through the C ABI): MultilinearPolynomial operations vs the CPU oracle, bit-exact.
Mirrors the reference's own tests in polynomials/src/multilinear/evaluation_form.rs:166-278."""
import ctypes as C
import random

import numpy as np
import pytest

import __graft_entry__ as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu

FIELDS = [0, 1, 2, 3]
FID = {"bn254_fq": 2, "bls12_381_fr": 0}


@pytest.fixture(scope="module")
def zk():
    zk = G.import_package()
    from zkmle_amd import _lib
    _lib.check(zk.lib().zk_init(0))
    return zk


def rand_table(zk, field, n, seed):
    t = np.zeros((n, zk.limbs(field)), np.uint64)
    rc = zk.lib().zk_host_fill_random(field, seed, 0, n, t.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert rc == 0
    return t


# ---- the reference's known answers, through the product path ------------------------------------
def test_reference_partial_evaluate_kats(zk, ref_kats):
    MP = zk.MultilinearPolynomial
    for k in ref_kats["partial_evaluate"]:
        f = FID[k["field"]]
        got = MP.partial_evaluate(MP.from_ints(f, k["poly"]), k["var"], zk.from_ints(f, [k["value"]]))
        assert got.to_ints() == k["expect"], k["src"]
    for k in ref_kats["evaluate"]:
        f = FID[k["field"]]
        assert zk.to_ints(f, MP.from_ints(f, k["poly"]).evaluate(zk.from_ints(f, k["values"]))) == [k["expect"]]
    for k in ref_kats["tensor_add"]:
        f = FID[k["field"]]
        assert MP.polynomial_tensor_add(MP.from_ints(f, k["wb"]), MP.from_ints(f, k["wc"])).to_ints() == k["expect"]
    for k in ref_kats["tensor_mul"]:
        f = FID[k["field"]]
        assert MP.polynomial_tensor_mul(MP.from_ints(f, k["wb"]), MP.from_ints(f, k["wc"])).to_ints() == k["expect"]


def test_reference_panics(zk, ref_kats):
    MP = zk.MultilinearPolynomial
    with pytest.raises(zk.ReferencePanic, match="power of 2"):         # evaluation_form.rs:171-176
        MP.from_ints(2, [0, 0, 3, 8, 0, 0])
    for k in ref_kats["tensor_mul_panics"]:                              # :269-277
        with pytest.raises(zk.ReferencePanic, match="Different polynomial length"):
            MP.polynomial_tensor_mul(MP.from_ints(2, k["wb"]), MP.from_ints(2, k["wc"]))
    t = MP.from_ints(2, [0, 0, 3, 8])
    with pytest.raises(zk.ReferencePanic):                               # too many values: empty Vec -> :13
        t.evaluate(zk.from_ints(2, [6, 2, 9]))
    with pytest.raises(zk.ReferencePanic):                               # variable out of range: :80 underflow
        MP.partial_evaluate(t, 2, zk.from_ints(2, [1]))
    with pytest.raises(zk.ReferencePanic):
        MP.add_polynomials(t, MP.from_ints(2, [1, 2]))                   # :149-153
    assert zk.to_ints(2, t.evaluate(zk.from_ints(2, [6]))) == [18]       # fewer values: element 0 of the rest


# ---- randomized parity vs the oracle ---------------------------------------------------------------
@pytest.mark.parametrize("field", FIELDS)
def test_fold_every_variable_small(zk, field):
    MP = zk.MultilinearPolynomial
    for n in range(1, 9):
        tab = rand_table(zk, field, 1 << n, 100 + n)
        r = rand_table(zk, field, 1, 999 + n)[0]
        poly = MP(field, tab)
        for var in range(n):
            got = MP.partial_evaluate(poly, var, r).evaluated_values
            assert np.array_equal(got, O.partial_evaluate(field, tab, var, r)), (field, n, var)


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("logn", [10, 13, 16])
def test_fold_var0_and_inner_vars_medium(zk, field, logn):
    MP = zk.MultilinearPolynomial
    tab = rand_table(zk, field, 1 << logn, 7 * logn + field)
    r = rand_table(zk, field, 1, 31337)[0]
    poly = MP(field, tab)
    for var in (0, 1, logn // 2, logn - 1):
        got = MP.partial_evaluate(poly, var, r).evaluated_values
        assert np.array_equal(got, O.partial_evaluate(field, tab, var, r)), (field, logn, var)


@pytest.mark.parametrize("field", [0, 2])
def test_edge_values(zk, field):
    """0, 1, p-1 and r in {0, 1, p-1}: exercises the conditional subtractions"""
    MP = zk.MultilinearPolynomial
    p = O.modulus(field)
    vals = [0, 1, p - 1, p - 2, 2, (p - 1) // 2, (p + 1) // 2, 0]
    tab = zk.from_ints(field, vals)
    for rv in (0, 1, p - 1, 2):
        r = zk.from_ints(field, [rv])[0]
        for var in range(3):
            got = MP.partial_evaluate(MP(field, tab), var, r).evaluated_values
            assert np.array_equal(got, O.partial_evaluate(field, tab, var, r))
    # fold by 0 / 1 selects the halves
    poly = MP(field, tab)
    assert MP.partial_evaluate(poly, 0, zk.from_ints(field, [0])[0]).to_ints() == [v % p for v in vals[:4]]
    assert MP.partial_evaluate(poly, 0, zk.from_ints(field, [1])[0]).to_ints() == [v % p for v in vals[4:]]


@pytest.mark.parametrize("field", FIELDS)
def test_evaluate_sums_bytes_and_elementwise(zk, field):
    MP = zk.MultilinearPolynomial
    logn = 12
    tab = rand_table(zk, field, 1 << logn, 555 + field)
    poly = MP(field, tab)
    point = rand_table(zk, field, logn, 777)
    assert np.array_equal(poly.evaluate(point), O.evaluate(field, tab, point))
    assert np.array_equal(poly.evaluate(point[:5]), O.evaluate(field, tab, point[:5]))
    assert np.array_equal(poly.evaluate(point[:0]), tab[0])
    assert np.array_equal(poly.sum(), O.vec_sum(field, tab))
    assert np.array_equal(poly.half_sums(), O.split_and_sum(field, tab))
    assert poly.convert_to_bytes() == O.mle_to_bytes(field, tab)
    s = point[3]
    assert np.array_equal(poly.scalar_mul(s).evaluated_values, O.scalar_mul(field, tab, s))
    other = rand_table(zk, field, 1 << logn, 4242)
    assert np.array_equal(MP.add_polynomials(poly, MP(field, other)).evaluated_values, O.add_polynomials(field, tab, other))
    neg = O.fe_op(field, "neg", s)
    assert np.array_equal(poly.sub_scalar(s).evaluated_values,
                          O.add_polynomials(field, tab, np.tile(neg, (1 << logn, 1))))
    w = tab[:32]
    assert np.array_equal(MP.polynomial_tensor_add(MP(field, w), MP(field, w)).evaluated_values, O.polynomial_tensor_add(field, w, w))
    assert np.array_equal(MP.polynomial_tensor_mul(MP(field, w), MP(field, w)).evaluated_values, O.polynomial_tensor_mul(field, w, w))
    assert poly.number_of_variables() == logn


@pytest.mark.parametrize("field", [0, 2])
@pytest.mark.parametrize("logn", [2, 3, 9, 14])
def test_fused_fold_half_sums(zk, field, logn):
    MP = zk.MultilinearPolynomial
    tab = rand_table(zk, field, 1 << logn, 900 + logn)
    r = rand_table(zk, field, 1, 17)[0]
    folded, sums = MP(field, tab).fold_half_sums(r)
    want = O.partial_evaluate(field, tab, 0, r)
    assert np.array_equal(folded.evaluated_values, want)
    assert np.array_equal(sums, O.split_and_sum(field, want))


def test_device_generator_matches_host_mirror(zk):
    for field in FIELDS:
        t = zk.MultilinearPolynomial.random(field, 1 << 10, 0xABCDEF)
        assert np.array_equal(t.evaluated_values, rand_table(zk, field, 1 << 10, 0xABCDEF))


def test_stateless_host_buffer_calls(zk):
    field = 0
    tab = rand_table(zk, field, 1 << 8, 3)
    r = rand_table(zk, field, 1, 4)[0]
    out = np.zeros((1 << 7, 4), np.uint64)
    u64p = C.POINTER(C.c_uint64)
    assert zk.lib().zk_host_partial_evaluate(field, tab.ctypes.data_as(u64p), 1 << 8, 0, r.ctypes.data_as(u64p), out.ctypes.data_as(u64p)) == 0
    assert np.array_equal(out, O.partial_evaluate(field, tab, 0, r))
    pt = rand_table(zk, field, 8, 5)
    ev = np.zeros(4, np.uint64)
    assert zk.lib().zk_host_evaluate(field, tab.ctypes.data_as(u64p), 1 << 8, pt.ctypes.data_as(u64p), 8, ev.ctypes.data_as(u64p)) == 0
    assert np.array_equal(ev, O.evaluate(field, tab, pt))


# ---- BASELINE sizes: the WHOLE output table against the oracle + size-independent properties ---------
@pytest.mark.parametrize("logn", [20, 24])
def test_full_size_fold_properties(zk, logn):
    field = 0
    MP = zk.MultilinearPolynomial
    n = 1 << logn
    poly = MP.random(field, n, 0x5EED0000 + logn)
    r = rand_table(zk, field, 1, 0x5EED0000 + n)[0]
    folded = MP.partial_evaluate(poly, 0, r)
    out = folded.evaluated_values
    # (1) sampled entries against the oracle's scalar formula y1 + r (y2 - y1)
    rng = random.Random(logn)
    idx = [0, 1, n // 2 - 1, n // 4] + [rng.randrange(n // 2) for _ in range(2000)]
    lo = np.zeros((len(idx), 4), np.uint64)
    hi = np.zeros((len(idx), 4), np.uint64)
    L = zk.lib()
    u64p = C.POINTER(C.c_uint64)
    for k, i in enumerate(idx):
        assert L.zk_host_fill_random(field, 0x5EED0000 + logn, i, 1, lo[k].ctypes.data_as(u64p)) == 0
        assert L.zk_host_fill_random(field, 0x5EED0000 + logn, i + n // 2, 1, hi[k].ctypes.data_as(u64p)) == 0
    pairs = np.concatenate([lo, hi])            # a 2*len(idx)-entry table whose var-0 fold is the sampled outputs
    m = 1
    while m < len(idx):
        m *= 2
    padded = np.zeros((2 * m, 4), np.uint64)
    padded[: len(idx)] = lo
    padded[m: m + len(idx)] = hi
    want = O.partial_evaluate(field, padded, 0, r)[: len(idx)]
    assert np.array_equal(out[idx], want)
    # (1b) every element: the oracle folds the host mirror of the whole table (0.5 s of CPU at 2^24)
    host = np.zeros((n, 4), np.uint64)
    assert L.zk_host_fill_random(field, 0x5EED0000 + logn, 0, n, host.ctypes.data_as(u64p)) == 0
    assert np.array_equal(poly.evaluated_values, host)          # device generator == host mirror
    want_full = O.partial_evaluate(field, host, 0, r)
    del host
    assert np.array_equal(out, want_full)
    # (2) linearity checksum: sum(fold(t, r)) == (1 - r) * sum(lo) + r * sum(hi)
    hs = poly.half_sums()
    one = O.from_ints(field, [1])[0]
    lhs = folded.sum()
    rhs = O.fe_op(field, "add", O.fe_op(field, "mul", O.fe_op(field, "sub", one, r), hs[0]), O.fe_op(field, "mul", r, hs[1]))
    assert np.array_equal(lhs, rhs)
    # (3) the fused round kernel gives the same table and the same half sums
    fused, sums = poly.fold_half_sums(r)
    assert np.array_equal(fused.evaluated_values, want_full)     # whole table of the fused round vs the oracle
    assert np.array_equal(sums, O.split_and_sum(field, want_full))
    assert np.array_equal(sums, folded.half_sums())
    del want_full
    # (4) evaluate == chained folds: f(r, x2..xn) evaluated two ways
    point = rand_table(zk, field, logn, 99)
    point[0] = r
    assert np.array_equal(poly.evaluate(point), folded.evaluate(point[1:]))


@pytest.mark.parametrize("field,logn", [(0, 16), (0, 17), (0, 19), (0, 20), (1, 16), (1, 18), (2, 17), (3, 18)])
def test_evaluate_large_vs_oracle(zk, field, logn):
    """evaluate (evaluation_form.rs:21-33) on tables large enough for the several-variables-per-pass kernel (zkmle_core.hip): full points
    and every kind of shorter point (fewer values: element 0 of what is left)"""
    MP = zk.MultilinearPolynomial
    tab = rand_table(zk, field, 1 << logn, 9000 + logn)
    poly = MP(field, tab)
    point = rand_table(zk, field, logn, 9100 + logn)
    for nv in (logn, logn - 1, logn - 3, 7, 5, 4, 3, 2, 1):
        assert np.array_equal(poly.evaluate(point[:nv]), O.evaluate(field, tab, point[:nv])), (field, logn, nv)
    assert np.array_equal(poly.evaluated_values, tab)            # the input is never folded in place
