"""End-to-end provers at BASELINE sizes (configs 2 and 5): basic sumcheck on a 2^n random table, and the
GKR sumcheck on 4 tables.  Splits the time into the whole-table transcript absorb and the rounds."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
for lg in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20,24").split(",")]:
    n = 1 << lg
    poly = zk.MultilinearPolynomial.random(0, n, 0x5EED0002)
    zk.Prover.init(0, poly).prove()                   # warm-up (a Prover proves once: prove() appends to its own transcript, prover.rs:10)
    prover = zk.Prover.init(0, poly)
    t0 = time.time(); proof = prover.prove(); t_prove = time.time() - t0
    st = zk.sumcheck.last_stats()
    t0 = time.time(); ok = zk.Verifier.init().verify(proof); t_verify = time.time() - t0
    # rounds only: the same fused kernels without the table absorb
    cur = poly
    t0 = time.time()
    sums = cur.half_sums()
    r = proof.round_univariate_polynomials[0][0]
    while len(cur) >= 4:
        cur, sums = cur.fold_half_sums(r)
    t_rounds = time.time() - t0
    print(json.dumps({"prover": "basic sumcheck (Prover::prove)", "log_n": lg, "prove_s": t_prove, "rounds_only_s": t_rounds,
                      "absorb_and_rest_s": t_prove - t_rounds,
                      "prove_absorb_ms": st["ms_absorb"], "prove_rounds_ms": st["ms_rounds"], "verify_s": t_verify, "verified": bool(ok),
                      "field_mul_per_s_rounds": (n - 1) / t_rounds}), flush=True)
    if lg <= 22:
      for lgg in sorted({12, 16, lg}):
        n = 1 << lgg
        tabs = [[zk.MultilinearPolynomial.random(0, n, 10 * p + f) for f in range(2)] for p in range(2)]
        sp = zk.SumPolynomial([zk.ProductPolynomial(t) for t in tabs])
        claimed = sp.add_polynomials_element_wise().sum()
        zk.sumcheck.prove(sp, claimed, zk.Transcript())
        t0 = time.time(); res = zk.sumcheck.prove(sp, claimed, zk.Transcript()); t_g = time.time() - t0
        v = zk.sumcheck.verify(res, zk.Transcript(), 0)
        print(json.dumps({"prover": "GKR sumcheck (4 tables)", "log_n": lgg, "prove_s": t_g, "rounds_ms": zk.sumcheck.last_stats()["ms_rounds"],
                          "verified": bool(v.is_proof_valid), "field_mul_per_s": 5 * 2 * n / t_g}), flush=True)
