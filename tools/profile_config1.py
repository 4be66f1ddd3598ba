"""BASELINE config 1 on the GPU, step by step (host wall time per step, median of 30): where 0.6 ms go on a 2^12-entry table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = 1 << lg
tab = np.zeros((n, 4), np.uint64)
_lib.check(zk.lib().zk_host_fill_random(0, 0x5EED0001, 0, n, _lib.p64(tab)))
zk.Verifier.init().verify(zk.Prover.init(0, tab).prove())
steps = {"upload": [], "init_sum": [], "prove": [], "verify": [], "total": []}
for _ in range(30):
    t0 = time.perf_counter()
    poly = zk.MultilinearPolynomial(0, tab)
    t1 = time.perf_counter()
    pr = zk.Prover.init(0, poly)
    t2 = time.perf_counter()
    proof = pr.prove()
    t3 = time.perf_counter()
    ok = zk.Verifier.init().verify(proof)
    t4 = time.perf_counter()
    for k, v in zip(steps, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)):
        steps[k].append(v * 1e6)
print({k: round(sorted(v)[len(v) // 2], 1) for k, v in steps.items()}, "us; verified", ok, flush=True)
os.environ["ZK_PROOF_TRACE"] = "1"
zk.Prover.init(0, zk.MultilinearPolynomial(0, tab)).prove()
