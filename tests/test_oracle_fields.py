"""Oracle pin 1: field constants, field axioms vs Python ints, Keccak KATs."""
import random

import numpy as np
import pytest

from oracle import oracle as O
from oracle import pymodel as M

APPENDIX_A = {
    O.FR381: (0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
              0x1824b159acc5056f998c4fefecbc4ff55884b7fa0003480200000001fffffffe,
              0x0748d9d99f59ff1105d314967254398f2b6cedcb87925c23c999e990f3f29c6d, 0xfffffffeffffffff),
    O.FQ381: (0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
              0x15f65ec3fa80e4935c071a97a256ec6d77ce5853705257455f48985753c758baebf4000bc40c0002760900000002fffd,
              0x11988fe592cae3aa9a793e85b519952d67eb88a9939d83c08de5476c4c95b6d50a76e6a609d104f1f4df1f341c341746,
              0x89f3fffcfffcfffd),
    O.BN254_FQ: (0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,
                 0x0e0a77c19a07df2f666ea36f7879462c0a78eb28f5c70b3dd35d438dc58f0d9d,
                 0x06d89f71cab8351f47ab1eff0a417ff6b5e71911d44501fbf32cfc5b538afa89, 0x87d20782e4866389),
    O.BN254_FR: (0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001,
                 0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb,
                 0x0216d0b17f4e44a58c49833d53bb808553fe3ab1e35c59e31bb8e645ae216da7, 0xc2e1f593efffffff),
}


@pytest.mark.parametrize("field", list(APPENDIX_A))
def test_constants_match_appendix_a(field):
    assert O.constants(field) == APPENDIX_A[field]
    p, r, r2, inv = O.constants(field)
    nbits = 64 * O.limbs(field)
    assert r == (1 << nbits) % p and r2 == pow(1 << nbits, 2, p)
    assert (p * inv + 1) % (1 << 64) == 0
    assert p == M.P[O.FIELD_NAMES[field]]


@pytest.mark.parametrize("field", list(APPENDIX_A))
def test_field_ops_match_python_ints(field):
    rng = random.Random(1234 + field)
    p = O.modulus(field)
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << 64) - 1, 1 << 64, (1 << 128) + 5]
    vals = edge + [rng.randrange(p) for _ in range(40)]
    mont = O.from_ints(field, vals)
    assert O.to_ints(field, mont) == [v % p for v in vals]
    for _ in range(200):
        i, j = rng.randrange(len(vals)), rng.randrange(len(vals))
        a, b = vals[i] % p, vals[j] % p
        assert O.to_ints(field, O.fe_op(field, "add", mont[i], mont[j]))[0] == (a + b) % p
        assert O.to_ints(field, O.fe_op(field, "sub", mont[i], mont[j]))[0] == (a - b) % p
        assert O.to_ints(field, O.fe_op(field, "mul", mont[i], mont[j]))[0] == (a * b) % p
        assert O.to_ints(field, O.fe_op(field, "neg", mont[i]))[0] == (-a) % p
        if a:
            assert O.to_ints(field, O.fe_op(field, "inv", mont[i]))[0] == pow(a, -1, p)


@pytest.mark.parametrize("field", list(APPENDIX_A))
def test_montgomery_layout_is_arkworks(field):
    """in-memory limbs = (v * R) mod p, little-endian u64 [ark-ff layout, SURVEY 8(b)]"""
    p, r, _, _ = O.constants(field)
    for v in (0, 1, 5, p - 1, 0x123456789abcdef0123456789abcdef):
        assert O.limbs_to_int(O.from_ints(field, [v])[0]) == (v % p) * r % p


@pytest.mark.parametrize("field", list(APPENDIX_A))
def test_from_le_bytes_mod_order_and_byte_orders(field):
    rng = random.Random(77)
    p = O.modulus(field)
    for n in (0, 1, 31, 32, 33, 48, 64):
        data = bytes(rng.randrange(256) for _ in range(n))
        assert O.to_ints(field, O.from_le_bytes_mod_order(field, data))[0] == int.from_bytes(data, "little") % p
    assert O.to_ints(field, O.from_le_bytes_mod_order(field, b"\xff" * 32))[0] == (2 ** 256 - 1) % p
    nb = 8 * O.limbs(field)
    v = rng.randrange(p)
    m = O.from_ints(field, [v])[0]
    assert O.fe_to_bytes_be(field, m) == v.to_bytes(nb, "big")
    assert O.fe_to_bytes_le(field, m) == v.to_bytes(nb, "little")


def test_keccak_kats(derived_kats):
    for kat in derived_kats["keccak256"]:
        msg = bytes.fromhex(kat["msg_hex"])
        assert O.keccak256(msg).hex() == kat["digest"]
        assert M.keccak256(msg).hex() == kat["digest"]


def test_keccak_block_boundaries_c_vs_python():
    rng = random.Random(5)
    for n in (1, 55, 135, 136, 137, 271, 272, 273, 500):
        data = bytes(rng.randrange(256) for _ in range(n))
        assert O.keccak256(data) == M.keccak256(data)


def test_transcript_model(derived_kats):
    """fiat_shamir_transcript.rs:51-62 scenario; digest absorbed back, running state kept (:29-36)"""
    kat = derived_kats["transcript"]
    t, m = O.Transcript(), M.Transcript()
    t.append(kat["append"].encode())
    m.append(kat["append"].encode())
    first = t.sample_random_challenge()
    assert first.hex() == kat["first_sample"] == m.sample().hex()
    c = O.to_ints(O.BN254_FQ, t.random_challenge_as_field_element(O.BN254_FQ))[0]
    assert c == int(kat["then_challenge_bn254_fq"], 16) == m.challenge(M.P["bn254_fq"])
    # equivalent closed form: challenge_k = keccak(all appended bytes || all earlier digests)
    assert first == O.keccak256(b"boy")
    second = O.keccak256(b"boy" + first)
    assert int.from_bytes(second, "little") % M.P["bn254_fq"] == c


def test_transcript_incremental_equals_oneshot():
    rng = random.Random(9)
    t = O.Transcript()
    whole = b""
    for n in (3, 200, 0, 136, 1, 135, 137):
        chunk = bytes(rng.randrange(256) for _ in range(n))
        t.append(chunk)
        whole += chunk
    assert t.sample_random_challenge() == O.keccak256(whole)
