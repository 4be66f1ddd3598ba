"""BASELINE config 4: GKR prover, depth-3 layered circuit, 2^log_gates gates per layer, 1 x MI355X.
Sparse (linear-time) prover; verified with the sparse verifier.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wiring = sys.argv[3] if len(sys.argv) > 3 else "random"
field = 0
rng = np.random.default_rng(0x5EED0004)
n = 1 << lg
rows, out_bits = [], []
for l in range(depth):
    g = np.zeros((n, 4), np.uint64)
    if wiring == "random":
        g[:, 0] = rng.integers(0, n, n); g[:, 1] = rng.integers(0, n, n)
    else:                                      # regular fan-in-2 wiring on a layer of the same width (wraps)
        g[:, 0] = (2 * np.arange(n)) % n; g[:, 1] = (2 * np.arange(n) + 1) % n
    g[:, 2] = np.arange(n); g[:, 3] = rng.integers(0, 2, n)
    rows.append(g); out_bits.append(lg)
x = zk.MultilinearPolynomial.random(field, n, 0x5EED0004).evaluated_values
zk.gkr.sparse_prove(field, [r[: 1 << 10] % (1 << 10) for r in rows], [10] * depth, x[: 1 << 10])     # warm-up (small)
t0 = time.time()
circuit = zk.gkr.SparseCircuit(rows, out_bits, n)          # once per circuit: upload + grouping of the gate lists
t_compile = time.time() - t0
t0 = time.time()
proof = zk.gkr.sparse_prove(field, None, None, x, circuit=circuit)
t_prove = time.time() - t0
t0 = time.time()
ok = zk.gkr.sparse_verify(field, rows, out_bits, proof, x)
t_verify = time.time() - t0
k = lg
# field multiplications in the fused round kernels per layer: phase 1 + phase 2, 4 tables of 2^k: 3.5 * 2^k per first round, geometric
muls_rounds = depth * 2 * sum(3.5 * (1 << m) for m in range(2, k + 1))
print(json.dumps({"config": f"GKR prover, depth-{depth} layered circuit, 2^{lg} gates/layer ({wiring} wiring), BLS12-381 Fr, 1xMI355X",
                  "circuit_compile_s": t_compile, "prove_s": t_prove, "device_ms_per_layer": proof.ms_layers, "verify_s": t_verify, "verified": bool(ok),
                  "gates_per_s": depth * n / t_prove, "round_kernel_field_muls": muls_rounds,
                  "note": "prove_s includes the Keccak absorb of the 2^%d-entry output layer (sequential, host); circuit_compile_s is paid once per circuit" % lg}), flush=True)
