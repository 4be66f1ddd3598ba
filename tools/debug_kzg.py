import sys, os, time, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(40, exit=True)
import numpy as np
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
from oracle import oracle as O
log = open("gpurun_out/debug_kzg.log", "w")
def P(*a):
    print(*a, file=log, flush=True); print(*a, flush=True)
_lib.check(zk.lib().zk_init(0))
t=time.time(); P("start")
taus = zk.from_ints(0, [5, 2, 3])
lb = zk.kzg.compute_lagrange_basis(taus); P("lagrange", time.time()-t, lb.to_ints()[:3])
setup = zk.TrustedSetup.initialize_setup(taus); P("setup", time.time()-t)
pts = setup.g1_powers_of_tau.points(); P("points", time.time()-t, np.array_equal(pts, O.kzg_setup_g1(taus)))
vals = zk.from_ints(0, [0,4,0,4,0,4,3,7])
poly = zk.MultilinearPolynomial(0, vals)
c, st = zk.kzg.msm(poly, setup.g1_powers_of_tau, 0, True); P("msm", time.time()-t, st, np.array_equal(c, O.kzg_commit(vals, pts)))
key = setup.opening_key(); P("key", time.time()-t)
proof = zk.MultilinearKZG.open_and_prove(poly, setup, zk.from_ints(0,[6,4,0])); P("open", time.time()-t)
ev, proofs = O.kzg_open(vals, pts, zk.from_ints(0,[6,4,0]))
P("open ok", np.array_equal(proof.evaluation, ev), np.array_equal(proof.proofs, proofs))
