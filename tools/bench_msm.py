"""MSM timing at BASELINE sizes: prints one JSON line per (n, window) with the phase breakdown.
    python tools/bench_msm.py 22,24 16,18,19,20,21,22 [20,22,24]
third argument: window sizes to run on precomputed window-shifted bases (zk_g1_bases_precompute) as well."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
logs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,20").split(",")]
wins = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
pre = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else []
a = zk.from_ints(0, [0x1234567])[0]
d = zk.from_ints(0, [0x9abcdef12345])[0]


def run(scalars, bases, c, lg, extra):
    zk.lib().zk_release_cached_memory()                 # every window size starts from an empty scratch pool (its first call fills it)
    best, ref = None, None
    for rep in range(4):
        t0 = time.time()
        out, st = zk.kzg.msm(scalars, bases, c, True)
        wall = time.time() - t0
        if rep and (best is None or st["ms_total"] < best[0]["ms_total"]):
            best = (st, wall)
    st, wall = best
    n = 1 << lg
    st.update(log_n=lg, wall_ms=wall * 1e3, g1_add_per_s=st["windows"] * n / (st["ms_total"] * 1e-3), terms_per_s=n / (st["ms_total"] * 1e-3),
              point_x_limb0=int(out[0]), **extra)
    print(json.dumps(st), flush=True)
    return out


for lg in logs:
    n = 1 << lg
    bases = zk.G1Bases.synthetic(n, a, d)
    scalars = zk.MultilinearPolynomial.random(0, n, 0x5EED0003)
    zk.lib().zk_device_synchronize()
    ref = None
    for c in wins:
        out = run(scalars, bases, c, lg, {"precomputed": False})
        assert ref is None or np.array_equal(out, ref), "window sizes disagree"
        ref = out
    for c in pre:
        t0 = time.time()
        used = bases.precompute(c)
        build_s = time.time() - t0
        out = run(scalars, bases, used, lg, {"precomputed": True, "precompute_s": build_s, "table_bytes": 128 * n * ((256 + used - 1) // used)})
        assert ref is None or np.array_equal(out, ref), "precomputed bases disagree"
    del bases, scalars
    zk.lib().zk_release_cached_memory()
