"""Workload for rocprofv3: the GKR sumcheck (sumcheck_gkr_protocol.rs:24-67) on 4 tables of 2^log_n entries, `reps` proofs.
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 tools/profile_gkr_round.py 22 20
The first fused round of each proof is the largest dispatch of fold_round_evals_kernel (q = 2^(log_n - 2) pair indices)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << lg
tabs = [[zk.MultilinearPolynomial.random(0, n, 10 * p + f) for f in range(2)] for p in range(2)]
sp = zk.SumPolynomial([zk.ProductPolynomial(t) for t in tabs])
claimed = sp.add_polynomials_element_wise().sum()
for _ in range(3):
    zk.sumcheck.prove(sp, claimed, zk.Transcript())
t0 = time.time()
ms = []
for _ in range(reps):
    zk.sumcheck.prove(sp, claimed, zk.Transcript())
    ms.append(zk.sumcheck.last_stats()["ms_rounds"])
print({"log_n": lg, "reps": reps, "variant": os.environ.get("ZK_FRE_VARIANT", "0"), "ms_rounds_median": sorted(ms)[len(ms) // 2], "wall_ms_per_proof": (time.time() - t0) / reps * 1e3}, flush=True)
