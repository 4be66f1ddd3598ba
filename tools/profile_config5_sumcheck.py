"""Kernel trace target: config 5's sumcheck at one rank (zk_sharded_sumcheck_basic_prove, whole-table absorb off) on one 2^24 table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
S = zk.sharded
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
poly = zk.MultilinearPolynomial.random(0, 1 << lg, 0x5EED0002)
comm = S.Comm()
shard = S.GpuShard(poly)
for _ in range(3):
    S.sumcheck_basic_prove_device(comm, shard, absorb_table=False)
t0 = time.perf_counter()
for _ in range(10):
    S.sumcheck_basic_prove_device(comm, shard, absorb_table=False)
print({"ms_per_proof": (time.perf_counter() - t0) * 100})
if os.environ.get("ZK_PROOF_TRACE") == "1":
    import ctypes
    t0 = time.perf_counter()
    S.sumcheck_basic_prove_device(comm, shard, absorb_table=False)
    print({"python_call_us": (time.perf_counter() - t0) * 1e6})
