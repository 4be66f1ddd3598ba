"""Kernel trace target: the basic sumcheck prover on one 2^24 table (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
poly = zk.MultilinearPolynomial.random(0, 1 << 24, 0x5EED0002)
for _ in range(6):
    zk.Prover.init(0, poly).prove()
print(zk.sumcheck.last_stats())
