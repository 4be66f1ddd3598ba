import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
L = zk.lib()
for lg in (20, 24):
    n = 1 << lg
    t = zk.MultilinearPolynomial.random(0, n, 5)
    out = zk.MultilinearPolynomial.alloc(0, n // 2)
    r = t.evaluated_values[3] if lg <= 20 else zk.from_ints(0, [123456789])[0]
    sums = np.zeros((2, 4), np.uint64)
    def fused():
        _lib.check(L.zk_mle_fold_half_sums(t._h, _lib.p64(r), out._h, _lib.p64(sums), None))
    def hs():
        t.half_sums()
    def plain():
        _lib.check(L.zk_mle_fold(t._h, 0, _lib.p64(r), out._h, None))
    for name, fn, bytes_ in (("fold_half_sums (fused round)", fused, 48 * n), ("half_sums (read only)", hs, 32 * n), ("fold", plain, 48 * n)):
        for _ in range(200): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 500
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"2^{lg} {name}: {dt*1e6:.1f} us  {bytes_/dt/1e9:.0f} GB/s", flush=True)
