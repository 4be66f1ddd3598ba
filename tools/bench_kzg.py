"""Multilinear KZG end to end at BASELINE sizes (configs 3 and 5): trusted setup (Lagrange basis + 2^n fixed-base
scalar multiplications + batch normalisation), commit (one 2^n MSM), opening key (pre-summed bases), open_and_prove
(MSMs of 2^(n-1) ... 1 terms).  One JSON line per size."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))


def sync():
    zk.lib().zk_device_synchronize()


for lg in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,20").split(",")]:
    n = 1 << lg
    taus = zk.from_ints(0, [0x1000003 * (i + 1) + 12345 for i in range(lg)])
    point = zk.from_ints(0, [0x2000003 * (i + 7) + 999 for i in range(lg)])
    poly = zk.MultilinearPolynomial.random(0, n, 0x5EED0003)
    sync(); t0 = time.time()
    setup = zk.TrustedSetup.initialize_setup(taus)
    sync(); t_setup = time.time() - t0
    zk.MultilinearKZG.commit_to_polynomial(poly, setup)              # warm-up (scratch pool, base conversion)
    sync(); t0 = time.time()
    c = zk.MultilinearKZG.commit_to_polynomial(poly, setup)
    sync(); t_commit = time.time() - t0
    t0 = time.time()
    setup.opening_key()
    sync(); t_key = time.time() - t0
    zk.MultilinearKZG.open_and_prove(poly, setup, point)
    opens = []
    for _ in range(4):
        sync(); t0 = time.time()
        proof = zk.MultilinearKZG.open_and_prove(poly, setup, point)
        sync(); opens.append(time.time() - t0)
    t_open = min(opens)
    t0 = time.time()
    verified = bool(zk.MultilinearKZG.verify(setup, c, point, proof))          # :131-158, pairings on the host
    t_verify = time.time() - t0
    assert verified, "the opening that was timed does not verify"
    pre = {"verified": verified, "verify_s": t_verify}
    if len(sys.argv) > 2 and sys.argv[2] == "pre":          # the same on window-shifted copies of the setup and of the key's large levels
        t0 = time.time(); setup.precompute_for_commits(); setup.precompute_for_opens(); sync(); pre["precompute_s"] = time.time() - t0
        zk.MultilinearKZG.commit_to_polynomial(poly, setup)
        sync(); t0 = time.time(); c2 = zk.MultilinearKZG.commit_to_polynomial(poly, setup); sync(); pre["commit_pre_s"] = time.time() - t0
        zk.MultilinearKZG.open_and_prove(poly, setup, point)
        ts = []
        for _ in range(4):
            sync(); t0 = time.time(); p2 = zk.MultilinearKZG.open_and_prove(poly, setup, point); sync(); ts.append(time.time() - t0)
        pre["open_pre_s"] = min(ts)
        assert np.array_equal(c2, c) and np.array_equal(p2.proofs, proof.proofs)
    print(json.dumps({"log_n": lg, **pre, "setup_s": t_setup, "commit_s": t_commit, "opening_key_s": t_key, "open_s": t_open, "open_s_all": opens,
                      "commit_terms_per_s": n / t_commit, "open_terms_per_s": (n - 1) / t_open,
                      "note": "setup = compute_lagrange_basis + 2^n fixed-base [L_i(tau)]G + batch to affine; open = n MSMs of 2^(n-1)..1 terms on pre-summed bases"}), flush=True)
    del setup, proof
