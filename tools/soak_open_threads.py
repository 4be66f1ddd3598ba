"""Soak: several Python threads, each with its own stream, setup and opening key, run KZG openings (which themselves fan out to
worker threads / side streams) and GKR sumchecks concurrently; every opening is pairing-checked, every proof compared with a
single-threaded run.  One JSON line."""
import ctypes as C, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
L = zk.lib()
L.zk_set_stream.argtypes = [C.c_void_p]
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 21
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 3
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
errors, done = [], [0] * nthreads


def job(t):
    try:
        st = torch.cuda.Stream()
        assert L.zk_set_stream(C.c_void_p(st.cuda_stream)) == 0
        taus = zk.from_ints(0, [0x1000003 * (i + 1) + 77 * t + 5 for i in range(nv)])
        setup = zk.TrustedSetup.initialize_setup(taus)
        poly = zk.MultilinearPolynomial.random(0, 1 << nv, 900 + t)
        com = zk.MultilinearKZG.commit_to_polynomial(poly, setup)
        point = zk.from_ints(0, [0x2000003 * (i + 3) + t for i in range(nv)])
        first = None
        tabs = [[zk.MultilinearPolynomial.random(0, 1 << 14, 50 + 4 * t + 2 * p + f) for f in range(2)] for p in range(2)]
        S = zk.sumcheck
        sp = S.SumPolynomial([S.ProductPolynomial(pr) for pr in tabs])
        claimed = sp.add_polynomials_element_wise().sum()
        ref = None
        for it in range(iters):
            proof = zk.MultilinearKZG.open_and_prove(poly, setup, point)
            if first is None:
                first = proof
                assert zk.MultilinearKZG.verify(setup, com, point, proof) is True
            else:
                assert np.array_equal(first.proofs, proof.proofs) and np.array_equal(first.evaluation, proof.evaluation)
            r = S.prove(sp, claimed, zk.Transcript())
            if ref is None:
                ref = r
            else:
                assert np.array_equal(ref.round_univariate_polynomials, r.round_univariate_polynomials)
            done[t] += 1
    except Exception as e:          # noqa: BLE001
        errors.append((t, repr(e)))


t0 = time.time()
ths = [threading.Thread(target=job, args=(t,)) for t in range(nthreads)]
for th in ths:
    th.start()
for th in ths:
    th.join()
print(json.dumps({"log_n": nv, "threads": nthreads, "iterations_done": done, "errors": errors, "seconds": round(time.time() - t0, 2)}), flush=True)
sys.exit(1 if errors else 0)
