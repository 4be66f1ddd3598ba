import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
L = zk.lib()
for lg in (20, 24):
    n = 1 << lg
    tab = np.zeros((n, 4), np.uint64)
    _lib.check(L.zk_host_fill_random(0, 7, 0, n, _lib.p64(tab)))
    r = tab[5].copy()
    out = np.zeros((n // 2, 4), np.uint64)
    u64p = C.POINTER(C.c_uint64)
    L.zk_host_partial_evaluate(0, tab.ctypes.data_as(u64p), n, 0, r.ctypes.data_as(u64p), out.ctypes.data_as(u64p))
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        _lib.check(L.zk_host_partial_evaluate(0, tab.ctypes.data_as(u64p), n, 0, r.ctypes.data_as(u64p), out.ctypes.data_as(u64p)))
    dt = (time.perf_counter() - t0) / reps
    print(f"host-buffer partial_evaluate 2^{lg}: {dt*1e3:.2f} ms per call = {n/2/dt:.3e} field-mul/s (upload {32*n/2**20:.0f} MiB + fold + download {16*n/2**20:.0f} MiB, pageable host memory)", flush=True)
