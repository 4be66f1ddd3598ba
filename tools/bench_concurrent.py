"""Serving shape: many independent small proofs.  One thread on the default stream vs T threads, each on its own HIP stream
(zk_set_stream): small sumchecks are latency-bound (one workgroup's transcript step per round), so concurrent streams fill
the rest of the chip.  Prints proofs/s."""
import ctypes as C, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as G
zk = G.import_package()
from zkmle_amd import _lib
_lib.check(zk.lib().zk_init(0))
lib = zk.lib()
lib.zk_set_stream.argtypes = [C.c_void_p]
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
per_thread = 40


def make_job(seed):
    n = 1 << log_n
    tabs = [[zk.MultilinearPolynomial.random(0, n, seed + 2 * p + f) for f in range(2)] for p in range(2)]
    return zk.SumPolynomial([zk.ProductPolynomial(t) for t in tabs])


def run(sp, stream, count):
    if stream is not None:
        lib.zk_set_stream(C.c_void_p(stream.cuda_stream))
    claimed = np.zeros(4, np.uint64)
    for _ in range(count):
        zk.sumcheck.prove(sp, claimed, zk.Transcript())


for nthreads in (1, 2, 4, 8):
    jobs = [make_job(100 * t) for t in range(nthreads)]
    streams = [torch.cuda.Stream() if nthreads > 1 else None for _ in range(nthreads)]
    for sp, st in zip(jobs, streams):
        run(sp, st, 2)                                     # warm-up (main thread; restores nothing: each thread sets its own)
    lib.zk_set_stream(None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=run, args=(sp, st, per_thread)) for sp, st in zip(jobs, streams)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": f"GKR sumcheck, 4 tables of 2^{log_n}", "threads_and_streams": nthreads,
                      "proofs_per_s": nthreads * per_thread / dt, "ms_per_proof_per_thread": dt / per_thread * 1e3}), flush=True)
