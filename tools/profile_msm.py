"""Workload for rocprofv3: `reps` MSMs of 2^log_n terms at window size c (0 = default), optionally on precomputed bases.
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 tools/profile_msm.py 24 20 3 [pre]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
lg, c, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = 1 << lg
bases = zk.G1Bases.synthetic(n, zk.from_ints(0, [0x1234567])[0], zk.from_ints(0, [0x9abcdef12345])[0])
scalars = zk.MultilinearPolynomial.random(0, n, 0x5EED0003)
if len(sys.argv) > 4:
    c = bases.precompute(c)
for _ in range(reps):
    out, st = zk.kzg.msm(scalars, bases, c, True)
print(st)
