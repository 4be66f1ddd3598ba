import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
depth = 8
rng = random.Random(8)
rows = []
for i in range(depth):
    n_in = 1 << (i + 1)
    rows.append(np.array([[rng.randrange(n_in), rng.randrange(n_in), o, rng.choice([0, 1])] for o in range(1 << i)], np.uint64))
ob = [1] + list(range(1, depth))
x = zk.MultilinearPolynomial.random(0, 1 << depth, 0x5EED0008).evaluated_values
zk.gkr.sparse_prove(0, rows, ob, x)
circuit = zk.gkr.SparseCircuit(rows, ob, 1 << depth)
ts = []
for _ in range(7):
    t0 = time.perf_counter(); p = zk.gkr.sparse_prove(0, None, None, x, circuit=circuit); ts.append(time.perf_counter() - t0)
print({"sparse_ms_min": min(ts) * 1e3, "median": sorted(ts)[3] * 1e3, "device_ms_per_layer": p.ms_layers})
ts = []
for _ in range(5):
    t0 = time.perf_counter(); p = zk.gkr.sparse_prove(0, rows, ob, x); ts.append(time.perf_counter() - t0)
print({"sparse_uncompiled_ms_min": min(ts) * 1e3})
