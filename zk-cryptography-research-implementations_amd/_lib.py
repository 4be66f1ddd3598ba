"""Loads libzkmle_amd.so and declares the C ABI of include/zkmle.h.  Fails loudly if it is absent."""
import ctypes as C
import os

FR381, FQ381, BN254_FQ, BN254_FR = 0, 1, 2, 3
FIELD_NAMES = {FR381: "bls12_381_fr", FQ381: "bls12_381_fq", BN254_FQ: "bn254_fq", BN254_FR: "bn254_fr"}

ZK_OK = 0
ZK_E_NOT_POW2, ZK_E_LEN_MISMATCH, ZK_E_NVARS, ZK_E_NEED_TWO, ZK_E_KZG_LEN, ZK_E_RANGE = -1, -2, -3, -4, -5, -6
ZK_E_ARG, ZK_E_NOMEM, ZK_E_NO_DEVICE, ZK_E_HIP, ZK_E_NOT_INIT, ZK_E_COMM = -7, -8, -9, -10, -11, -12
_PANIC_CODES = {ZK_E_NOT_POW2, ZK_E_LEN_MISMATCH, ZK_E_NVARS, ZK_E_NEED_TWO, ZK_E_KZG_LEN, ZK_E_RANGE, ZK_E_NOT_INIT}

_HERE = os.path.dirname(os.path.abspath(__file__))


class ZkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[zk_status {code}] {msg}")
        self.code = code


class ReferencePanic(ZkError):
    """A precondition on which the reference panics (same message text)."""


def library_path():
    return os.environ.get("ZKMLE_AMD_LIB") or os.path.join(_HERE, "libzkmle_amd.so")


_lib = None
u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p
sz = C.c_size_t


def lib():
    """The loaded shared library.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ZkError(ZK_E_NO_DEVICE, f"{path} not found: build it with __graft_entry__.build() "
                                      "(zkmle_amd has no CPU fallback)")
    try:
        import torch  # noqa: F401  (loads the HIP runtime torch ships, so both share one libamdhip64)
    except Exception:
        pass
    L = C.CDLL(path)
    L.zk_status_message.restype = C.c_char_p
    L.zk_last_error.restype = C.c_char_p
    L.zk_version.restype = C.c_char_p
    L.zk_table_len.restype = sz
    L.zk_table_device_ptr.restype = vp
    sigs = {
        "zk_device_count": [C.POINTER(C.c_int)],
        "zk_init": [C.c_int],
        "zk_field_limbs": [C.c_int],
        "zk_device_synchronize": [],
        "zk_table_alloc": [C.c_int, sz, C.POINTER(vp)],
        "zk_table_upload": [C.c_int, u64p, sz, C.POINTER(vp)],
        "zk_table_upload_raw": [C.c_int, u64p, sz, C.POINTER(vp)],
        "zk_table_download": [vp, u64p],
        "zk_table_free": [vp],
        "zk_table_len": [vp],
        "zk_table_field": [vp],
        "zk_table_device_ptr": [vp],
        "zk_table_wrap": [C.c_int, vp, sz, C.POINTER(vp)],
        "zk_table_clone": [vp, C.POINTER(vp)],
        "zk_table_fill_random": [vp, C.c_uint64],
        "zk_table_fill_random_strided": [vp, C.c_uint64, sz, sz],
        "zk_host_fill_random": [C.c_int, C.c_uint64, sz, sz, u64p],
        "zk_mle_fold": [vp, sz, u64p, vp, vp],
        "zk_mle_fold_ptr": [C.c_int, vp, sz, sz, u64p, vp, vp],
        "zk_mle_evaluate": [vp, u64p, sz, u64p],
        "zk_mle_to_bytes": [vp, u8p],
        "zk_mle_scalar_mul": [vp, u64p, vp, vp],
        "zk_mle_add": [vp, vp, vp, vp],
        "zk_mle_sub_scalar": [vp, u64p, vp, vp],
        "zk_mle_tensor_add": [vp, vp, vp, vp],
        "zk_mle_tensor_mul": [vp, vp, vp, vp],
        "zk_mle_sum": [vp, u64p],
        "zk_mle_half_sums": [vp, u64p],
        "zk_mle_fold_half_sums": [vp, u64p, vp, u64p, vp],
        "zk_host_partial_evaluate": [C.c_int, u64p, sz, sz, u64p, u64p],
        "zk_host_evaluate": [C.c_int, u64p, sz, u64p, sz, u64p],
        "zk_fe_from_u64": [C.c_int, C.c_uint64, u64p],
        "zk_fe_to_bytes_be": [C.c_int, u64p, u8p],
        "zk_fe_from_le_bytes_mod_order": [C.c_int, u8p, sz, u64p],
        "zk_vec_from_canonical": [C.c_int, u64p, sz, u64p],
        "zk_vec_to_canonical": [C.c_int, u64p, sz, u64p],
    }
    for name, args in sigs.items():
        fn = getattr(L, name)       # AttributeError = missing export: loud
        fn.argtypes = args
        if name not in ("zk_table_len", "zk_table_device_ptr"):
            fn.restype = C.c_int
    _lib = L
    return L


def check(rc):
    if rc == ZK_OK:
        return
    L = lib()
    msg = L.zk_status_message(rc).decode()
    if rc == ZK_E_HIP or rc == ZK_E_NO_DEVICE or rc == ZK_E_COMM:
        msg += ": " + L.zk_last_error().decode()
    raise (ReferencePanic if rc in _PANIC_CODES else ZkError)(rc, msg)


def p64(arr):
    return arr.ctypes.data_as(u64p)


def p8(arr):
    return arr.ctypes.data_as(u8p)
