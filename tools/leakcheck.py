"""Repeated prover / MSM / KZG calls must not grow device memory: prints the free-memory delta after a warm-up."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20


def job():
    n = 1 << 16
    poly = zk.MultilinearPolynomial.random(0, n, 7)
    zk.Prover.init(0, poly).prove()
    tabs = [[zk.MultilinearPolynomial.random(0, n, 10 * p + f) for f in range(2)] for p in range(2)]
    sp = zk.SumPolynomial([zk.ProductPolynomial(t) for t in tabs])
    zk.sumcheck.prove(sp, np.zeros(4, np.uint64), zk.Transcript())
    taus = zk.from_ints(0, [3 + i for i in range(12)])
    setup = zk.TrustedSetup.initialize_setup(taus)
    p12 = zk.MultilinearPolynomial.random(0, 1 << 12, 9)
    c = zk.MultilinearKZG.commit_to_polynomial(p12, setup)
    pr = zk.MultilinearKZG.open_and_prove(p12, setup, taus)
    assert zk.MultilinearKZG.verify(setup, c, taus, pr)
    setup.precompute_for_commits()                       # window-shifted copies of the setup and of the opening key's levels: freed with their handles
    setup.precompute_for_opens(min_points=1 << 8)
    assert np.array_equal(zk.MultilinearKZG.open_and_prove(p12, setup, taus).proofs, pr.proofs)


for _ in range(3):
    job()
a = free_mb()
for _ in range(20):
    job()
b = free_mb()
print(f"free before {a:.0f} MiB, after 20 more rounds {b:.0f} MiB, delta {a - b:.1f} MiB", flush=True)
assert a - b < 64, "device memory grows with repeated calls"
