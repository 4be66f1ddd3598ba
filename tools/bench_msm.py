"""MSM timing at BASELINE sizes: prints one JSON line per (n, window) with the phase breakdown."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
logs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,20").split(",")]
wins = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
a = zk.from_ints(0, [0x1234567])[0]
d = zk.from_ints(0, [0x9abcdef12345])[0]
for lg in logs:
    n = 1 << lg
    t0 = time.time()
    bases = zk.G1Bases.synthetic(n, a, d)
    scalars = zk.MultilinearPolynomial.random(0, n, 0x5EED0003)
    zk.lib().zk_device_synchronize()
    tgen = time.time() - t0
    for c in wins:
        best = None
        for rep in range(3):
            t0 = time.time()
            out, st = zk.kzg.msm(scalars, bases, c, True)
            wall = time.time() - t0
            if best is None or st["ms_total"] < best[0]["ms_total"]:
                best = (st, wall)
        st, wall = best
        st.update(log_n=lg, wall_ms=wall * 1e3, gen_s=tgen, g1_add_per_s=st["windows"] * n / (st["ms_total"] * 1e-3),
                  terms_per_s=n / (st["ms_total"] * 1e-3))
        print(json.dumps(st), flush=True)
