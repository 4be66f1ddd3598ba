"""CPU-side checks of the product library: the C ABI loads, exports every symbol include/*.h
declares, refuses to compute without a GPU (no CPU fallback), and its tiny host-side helpers
(control path) agree with the oracle."""
import ctypes as C
import glob
import os
import random
import re

import numpy as np
import pytest

import __graft_entry__ as G
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def zk():
    G.build()
    return G.import_package()


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_every_declared_symbol_is_exported(zk):
    lib = C.CDLL(zk.library_path())
    syms = declared_symbols()
    assert len(syms) > 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_no_cpu_fallback_without_gpu(zk):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    n = C.c_int(-1)
    assert zk.lib().zk_device_count(C.byref(n)) == 0 and n.value == 0
    with pytest.raises(zk.ZkError) as e:
        zk.MultilinearPolynomial.from_ints(zk.FR381, [1, 2, 3, 4])
    assert e.value.code == -9            # ZK_E_NO_DEVICE, loud
    out = np.zeros(4, np.uint64)
    tab = zk.from_ints(zk.FR381, [1, 2, 3, 4])
    r = zk.from_ints(zk.FR381, [5])
    rc = zk.lib().zk_host_partial_evaluate(zk.FR381, tab.ctypes.data_as(C.POINTER(C.c_uint64)), 4, 0,
                                           r.ctypes.data_as(C.POINTER(C.c_uint64)),
                                           out.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert rc == -9


def test_precondition_codes_come_before_device_checks(zk):
    # a non-power-of-two upload is the reference's panic, GPU or not
    with pytest.raises(zk.ReferencePanic) as e:
        zk.MultilinearPolynomial.from_ints(zk.BN254_FQ, [0, 0, 3, 8, 0, 0])
    assert e.value.code == -1 and "power of 2" in str(e.value)
    assert zk.lib().zk_status_message(-3) == b"different number of variables"
    assert zk.lib().zk_status_message(-11) == b"Can't prove without init"


@pytest.mark.parametrize("field", [0, 1, 2, 3])
def test_host_helpers_match_oracle(zk, field):
    rng = random.Random(field)
    p = O.modulus(field)
    vals = [0, 1, p - 1, -5] + [rng.randrange(p) for _ in range(20)]
    assert np.array_equal(zk.from_ints(field, vals), O.from_ints(field, vals))
    assert zk.to_ints(field, O.from_ints(field, vals)) == [v % p for v in vals]
    L = zk.lib()
    n = zk.limbs(field)
    for v in (0, 7, 2 ** 64 - 1):
        out = np.zeros(n, np.uint64)
        assert L.zk_fe_from_u64(field, v, out.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
        assert np.array_equal(out, O.from_ints(field, [v])[0])
    for nb in (0, 1, 31, 32, 33, 64):
        data = bytes(rng.randrange(256) for _ in range(nb))
        buf = np.frombuffer(data, np.uint8).copy() if nb else np.zeros(1, np.uint8)
        out = np.zeros(n, np.uint64)
        assert L.zk_fe_from_le_bytes_mod_order(field, buf.ctypes.data_as(C.POINTER(C.c_uint8)), nb,
                                               out.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
        assert np.array_equal(out, O.from_le_bytes_mod_order(field, data))
    # the short-input path (a few subtractions + one product, csrc/host_field.h): multiples of p and their neighbours, full-width values
    width = 8 * n
    edge = [0, 1, p - 1, p, p + 1, 2 * p - 1, 2 * p, 2 * p + 1, (1 << (8 * width)) - 1, (1 << (8 * width)) - p]
    edge += [k * p + d for k in range(3, 16) for d in (-1, 0, 1) if 0 <= k * p + d < (1 << (8 * width))]
    for v in edge:
        for nb in (width, 32) if width != 32 else (32,):
            if v >= 1 << (8 * nb):
                continue
            data = v.to_bytes(nb, "little")
            buf = np.frombuffer(data, np.uint8).copy()
            out = np.zeros(n, np.uint64)
            assert L.zk_fe_from_le_bytes_mod_order(field, buf.ctypes.data_as(C.POINTER(C.c_uint8)), nb,
                                                   out.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
            assert zk.to_ints(field, out.reshape(1, -1)) == [v % p], (field, hex(v), nb)
    m = O.from_ints(field, [vals[5]])[0]
    be = np.zeros(8 * n, np.uint8)
    assert L.zk_fe_to_bytes_be(field, m.ctypes.data_as(C.POINTER(C.c_uint64)), be.ctypes.data_as(C.POINTER(C.c_uint8))) == 0
    assert be.tobytes() == O.fe_to_bytes_be(field, m)


@pytest.mark.parametrize("field", [0, 1, 2, 3])
def test_synthetic_generator_is_reduced_and_reproducible(zk, field):
    n = zk.limbs(field)
    a = np.zeros((64, n), np.uint64)
    b = np.zeros((32, n), np.uint64)
    L = zk.lib()
    assert L.zk_host_fill_random(field, 42, 0, 64, a.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    assert L.zk_host_fill_random(field, 42, 32, 32, b.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    assert np.array_equal(a[32:], b)                      # counter based: shards reproduce
    p = O.modulus(field)
    ints = [O.limbs_to_int(row) for row in a]
    assert all(0 <= v < p for v in ints) and len(set(ints)) == 64


def test_comm_handle_and_sharded_entry_points_without_gpu(zk):
    """the multi-GPU boundary loads on a CPU-only host: a one-rank communicator needs no transport; the provers refuse to compute
    without a device (no CPU fallback) and reject bad communicators before anything else"""
    import torch
    S = zk.sharded
    lib = S._declare_host()
    ops = S.HostOps()
    h = C.c_void_p()
    assert lib.zk_comm_from_host_ops(C.byref(ops), 1, 0, C.byref(h)) == 0
    lib.zk_comm_rank.argtypes = [C.c_void_p]
    lib.zk_comm_size.argtypes = [C.c_void_p]
    assert lib.zk_comm_rank(h) == 0 and lib.zk_comm_size(h) == 1 and lib.zk_comm_backend(h) == b"host-ops"
    rx, n = C.c_uint64(7), C.c_uint64(7)
    assert lib.zk_comm_stats(h, C.byref(rx), C.byref(n)) == 0 and rx.value == 0 and n.value == 0
    assert lib.zk_comm_from_host_ops(C.byref(ops), 2, 0, C.byref(C.c_void_p())) == -7     # two ranks need the four callbacks
    assert lib.zk_comm_from_host_ops(C.byref(ops), 1, 1, C.byref(C.c_void_p())) == -7     # rank out of range
    out = np.zeros(4, np.uint64)
    assert lib.zk_sharded_sumcheck_basic_prove(None, None, 0, out.ctypes.data_as(C.POINTER(C.c_uint64)),
                                               out.ctypes.data_as(C.POINTER(C.c_uint64)), None) == -7
    assert lib.zk_comm_free(h) == 0
    if not torch.cuda.is_available():
        uid = np.zeros(128, np.uint8)
        hh = C.c_void_p()
        assert lib.zk_comm_init_rccl(uid.ctypes.data_as(C.POINTER(C.c_uint8)), 1, 0, C.byref(hh)) == -9   # ZK_E_NO_DEVICE, before RCCL is opened
