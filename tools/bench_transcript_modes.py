"""Round phase of the provers with the transcript step on the host (mailbox, default) or on the device (ZK_HOST_TRANSCRIPT=0):
run once per mode (the switch is read once per process).  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
out = {"ZK_HOST_TRANSCRIPT": os.environ.get("ZK_HOST_TRANSCRIPT", "1 (default)")}
for lg in (12, 16, 20, 22):
    n = 1 << lg
    tabs = [[zk.MultilinearPolynomial.random(0, n, 10 * p + f) for f in range(2)] for p in range(2)]
    sp = zk.SumPolynomial([zk.ProductPolynomial(t) for t in tabs])
    claimed = sp.add_polynomials_element_wise().sum()
    for _ in range(3):
        zk.sumcheck.prove(sp, claimed, zk.Transcript())
    ms = []
    for _ in range(20):
        zk.sumcheck.prove(sp, claimed, zk.Transcript())
        ms.append(zk.sumcheck.last_stats()["ms_rounds"])
    out[f"gkr_sumcheck_4x2p{lg}_ms"] = sorted(ms)[len(ms) // 2]
    del tabs, sp
for lg in (20, 24):
    poly = zk.MultilinearPolynomial.random(0, 1 << lg, 0x5EED0002)
    zk.Prover.init(0, poly).prove()
    ms = []
    for _ in range(3):
        pr = zk.Prover.init(0, poly)
        pr.prove()
        ms.append(zk.sumcheck.last_stats()["ms_rounds"])
    out[f"basic_sumcheck_2p{lg}_rounds_ms"] = sorted(ms)[len(ms) // 2]
    del poly, pr
print(json.dumps(out), flush=True)
