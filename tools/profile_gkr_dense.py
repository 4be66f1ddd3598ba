"""Workload for rocprofv3: gkr_protocol::prove (gkr_protocol.rs:26-143) on a reference-shaped circuit of the given depth (layer i: 2^i gates reading
2^(i+1) wires), `reps` proofs; bench.py's `paths.gkr_dense` shape.
    rocprofv3 --kernel-trace --stats -d /tmp/p -- python3 tools/profile_gkr_dense.py 8 5"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = random.Random(8)
layers = []
for i in range(depth):
    n_in = 1 << (i + 1)
    layers.append(zk.gkr.Layer([zk.gkr.Gate(rng.randrange(n_in), rng.randrange(n_in), o, rng.choice([0, 1])) for o in range(1 << i)]))
circuit = zk.gkr.Circuit(0, layers)
x = zk.MultilinearPolynomial.random(0, 1 << depth, 0x5EED0008).evaluated_values
zk.gkr.prove(circuit, x)
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    gp = zk.gkr.prove(circuit, x)
    ts.append(time.perf_counter() - t0)
print({"depth": depth, "ms_min": min(ts) * 1e3, "ms_median": sorted(ts)[len(ts) // 2] * 1e3, "verified": bool(zk.gkr.verify(circuit, gp, x))}, flush=True)
tv = []
for _ in range(5):
    t0 = time.perf_counter()
    okv = zk.gkr.verify(circuit, gp, x)
    tv.append(time.perf_counter() - t0)
print({"verify_ms_min": min(tv) * 1e3, "accepted": bool(okv)}, flush=True)
