"""Race hunt: the same statements proved over and over, every proof compared byte for byte with the first one of its kind (the provers are
deterministic: any difference is a lost update or a stale read in the exchanges that run in the producers' last workgroups).
    python3 tools/stress_determinism.py [seconds=120]
One JSON line: proofs per kind, mismatches."""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
S = zk.sharded
MP = zk.MultilinearPolynomial


def digest(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


kinds = {}
comm = S.Comm()
for lg in (24, 22, 20, 17, 13):                               # basic sumcheck rounds (absorb off): every pass structure
    shard = S.GpuShard(MP.random(0, 1 << lg, 0x5EED0002 + lg))
    kinds[f"basic_2p{lg}"] = (lambda shard=shard: digest(*S.sumcheck_basic_prove_device(comm, shard, absorb_table=False)))
for lg in (22, 18, 14):                                       # GKR sumcheck, 2 products of 2 factors
    tabs = [[MP.random(0, 1 << lg, 0xA000 + 10 * lg + 2 * p + f) for f in range(2)] for p in range(2)]
    sp = zk.SumPolynomial([zk.ProductPolynomial(t) for t in tabs])
    claimed = np.zeros(4, np.uint64)

    def gkr(sp=sp, claimed=claimed):
        r = zk.sumcheck.prove(sp, claimed, zk.Transcript())
        return digest(r.round_univariate_polynomials, r.random_challenges)
    kinds[f"gkr_sumcheck_4x2p{lg}"] = gkr
rng = np.random.default_rng(0xD37)
lg, depth = 18, 2                                             # sparse GKR, half-table gate weights
n = 1 << lg
rows = []
for _ in range(depth):
    g = np.zeros((n, 4), np.uint64)
    g[:, 0] = rng.integers(0, n, n); g[:, 1] = rng.integers(0, n, n); g[:, 2] = np.arange(n); g[:, 3] = rng.integers(0, 2, n)
    rows.append(g)
x = MP.random(0, n, 0xD38).evaluated_values
circuit = zk.gkr.SparseCircuit(rows, [lg] * depth, n)


def sparse():
    p = zk.gkr.sparse_prove(0, None, None, x, circuit=circuit)
    return digest(p.claimed_sum, p.layer_claims, p.coeffs, p.challenges, p.wb_evals, p.wc_evals)


kinds["sparse_gkr_2x2p18"] = sparse
first, counts, bad = {}, {k: 0 for k in kinds}, []
t0 = time.time()
while time.time() - t0 < budget:
    for k, fn in kinds.items():
        reps = 200 if k.startswith("basic") or k.startswith("gkr_sumcheck") else 5
        for _ in range(reps):
            d = fn()
            counts[k] += 1
            if k not in first:
                first[k] = d
            elif d != first[k]:
                bad.append([k, counts[k]])
        if time.time() - t0 >= budget:
            break
print(json.dumps({"seconds": budget, "proofs": counts, "mismatches": bad}), flush=True)
