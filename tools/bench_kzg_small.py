import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
MP = zk.MultilinearPolynomial
depth = 8
taus = MP.random(0, 16, 9).evaluated_values[:depth]
t0 = time.perf_counter(); setup = zk.TrustedSetup.initialize_setup(taus); print("setup ms", (time.perf_counter() - t0) * 1e3)
poly = MP.random(0, 1 << depth, 5)
opening = MP.random(0, 16, 11).evaluated_values[:depth]
def tm(f, n=5):
    f(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = f(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), r
print("commit ms", tm(lambda: zk.MultilinearKZG.commit_to_polynomial(poly, setup))[0])
print("open ms", tm(lambda: zk.MultilinearKZG.open_and_prove(poly, setup, opening))[0])
for lg in (4, 8, 12, 16):
    taus = MP.random(0, 32, 9).evaluated_values[:lg]
    s2 = zk.TrustedSetup.initialize_setup(taus)
    p2 = MP.random(0, 1 << lg, 5)
    op = MP.random(0, 32, 11).evaluated_values[:lg]
    print(lg, "commit", tm(lambda: zk.MultilinearKZG.commit_to_polynomial(p2, s2))[0], "open", tm(lambda: zk.MultilinearKZG.open_and_prove(p2, s2, op))[0])
