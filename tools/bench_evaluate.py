"""MultilinearPolynomial::evaluate (evaluation_form.rs:21-33) on one table: device time per call.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
out = {}
for lg in (20, 24):
    poly = zk.MultilinearPolynomial.random(0, 1 << lg, 0x5EED0002)
    point = zk.MultilinearPolynomial.random(0, lg, 77).evaluated_values
    for _ in range(3):
        poly.evaluate(point)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        poly.evaluate(point)
        ts.append((time.perf_counter() - t0) * 1e3)
    out[f"evaluate_2p{lg}_ms"] = sorted(ts)[len(ts) // 2]
print(json.dumps(out), flush=True)
