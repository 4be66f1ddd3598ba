"""The two round kernels of the GKR sumcheck (sumcheck_gkr_protocol.rs:113-143) alone, on 4 tables of 2^log_n entries (2 products x 2
factors): HIP events over back-to-back enqueue-only launches (zk_sumpoly_round_evals / zk_sumpoly_fold_round_evals with NULL outputs).
    python tools/bench_round_kernels.py [log_n=22] [reps=200]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ng = 1 << lg
MP = zk.MultilinearPolynomial
lib = zk.sharded._declare_host()
sc = zk.sumcheck._decl()
tabs = [MP.random(0, ng, 0x5EED0440 + k) for k in range(4)]
outs = [MP.alloc(0, ng // 2) for _ in range(4)]
ta = (C.c_void_p * 4)(*[t._h for t in tabs])
oa = (C.c_void_p * 4)(*[t._h for t in outs])
r = np.zeros(4, np.uint64)
_lib.check(zk.lib().zk_host_fill_random(0, 5, 77, 1, _lib.p64(r)))


def ev(fn, n, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fre = ev(lambda: _lib.check(lib.zk_sumpoly_fold_round_evals(ta, oa, 2, 2, _lib.p64(r), None)), reps)
re_ = ev(lambda: _lib.check(sc.zk_sumpoly_round_evals(ta, 2, 2, None)), reps)
fb, rb = 4 * (ng + ng // 2) * 32.0, 4 * ng * 32.0
print(json.dumps({"log_n": lg, "fold_round_evals_us": fre, "fold_round_evals_GBps": fb / fre / 1e3, "fold_frac": fb / fre / 1e3 / 8000,
                  "round_evals_us": re_, "round_evals_GBps": rb / re_ / 1e3, "round_frac": rb / re_ / 1e3 / 8000}), flush=True)
