"""bench.py -- field-mul/s of the 2^24 MLE fold (BASELINE.json metric) on N MI355X GPUs.

A step = one `partial_evaluate(table, 0, r)` pass over one resident 2^24-entry BLS12-381 Fr
table per GPU (2^23 field multiplications, 96 algorithmic bytes each).  Weak scaling: every
rank folds its own low-bit shard; the data path has no collective.
Prints ONE JSON line on rank 0.  See DESIGN.md section 5 for the roofline accounting.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-msm", action="store_true", help="skip the secondary 2^log-n MSM measurement")
    ap.add_argument("--msm-reps", type=int, default=3)
    ap.add_argument("--rehearse", action="store_true",
                    help="debug: run the N>1 code path with every rank on cuda:0 over gloo (one-GPU boxes)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__ as G
    zk = G.import_package()
    from zkmle_amd import _lib

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: zkmle_amd has no CPU fallback")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    _lib.check(zk.lib().zk_init(local_rank))
    backend_note = None
    if world > 1:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                probe = torch.zeros(1, device="cuda")
                dist.all_reduce(probe)                      # the first collective builds the communicator: fail here, not mid-run
                torch.cuda.synchronize()
            except Exception as e:                          # noqa: BLE001  -- keep the headline line: the collectives go over gloo
                backend_note = f"nccl (RCCL) unavailable, collectives over gloo with host staging: {e!r}"[:300]
                try:
                    dist.destroy_process_group()
                except Exception:                           # noqa: BLE001
                    pass
                os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
                dist.init_process_group("gloo")
                args.rehearse = True                        # host-staged tensors from here on (each rank keeps its own GPU)

    field = zk.FR381
    n = 1 << args.log_n
    half = n // 2
    MP = zk.MultilinearPolynomial
    table = MP.random(field, n, 0x5EED0005 + rank)       # shard-wise on-device generation
    out = MP.alloc(field, half)
    r = np.zeros(4, np.uint64)
    _lib.check(zk.lib().zk_host_fill_random(field, 0x5EED0005, n, 1, _lib.p64(r)))
    L = zk.lib()
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        _lib.check(L.zk_mle_fold(table._h, 0, _lib.p64(r), out._h, stream))
        out_len_fix()

    def out_len_fix():
        pass

    # clock pre-warm (untimed, outside the W / K protocol): a cold MI355X needs a few hundred milliseconds of
    # load before it holds its sustained clock; without this a short K reads 20 % low
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.4:
        for _ in range(50):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps          # HIP events on the launch stream
    if world > 1:
        tt = torch.tensor([dt], device="cpu" if args.rehearse else "cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    muls = half * args.steps * world
    value = muls / dt
    algo_bytes = 96.0 * half                               # per launch: 2 x 32 B read + 32 B write per mul
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
    result = {
        "metric": f"field-mul/s (2^{args.log_n} MLE fold)", "value": value, "unit": "field-mul/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": f"{args.log_n}-variable MLE fold (partial_evaluate var 0), BLS12-381 Fr, "
                               f"2^{args.log_n}-entry table per GPU", "log_n": args.log_n, "field": "bls12_381_fr",
                   "arithmetic": "255-bit Montgomery field, 8 x u32 limbs in HBM, products as 29-bit-limb v_mad_u64_u32 scans",
                   "sharding": "low-bit shard per rank, no data-path collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                     "traffic": None, "kernel": "fold0_kernel<Fr381>", "kernel_ms": kern_ms,
                     "algorithmic_bytes_per_launch": algo_bytes},
    }
    pmc = os.path.join(ROOT, "profiles", "r1", "fold_2p24_pmc.json")
    if args.log_n == 24 and os.path.exists(pmc):            # PMC passes are separate rocprofv3 runs (committed summary)
        with open(pmc) as f:
            result["roofline"]["traffic"] = json.load(f)["hbm_bytes_per_launch"]
        result["roofline"]["traffic_source"] = "profiles/r1/fold_2p24_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH doubled per gfx950 note)"
    if backend_note:
        result["collectives"] = backend_note
    if world > 1:
        try:                                               # a failure in a secondary leg must not cost the headline line
            result["exchange"] = sumcheck_round_exchange(zk, table, out, r, world, local_rank, args.rehearse)
        except Exception as e:                             # noqa: BLE001
            result["exchange"] = {"error": repr(e)}
        try:
            result["sharded_sumcheck"] = sharded_sumcheck_leg(zk, rank, world, local_rank, args.rehearse)
        except Exception as e:                             # noqa: BLE001
            result["sharded_sumcheck"] = {"error": repr(e)}
    if not args.no_msm:
        try:
            result["msm"] = msm_leg(zk, args, rank, world, local_rank)
            if rank == 0 and not args.no_cpu_baseline:
                result["msm"]["cpu_baseline"] = cpu_baseline_msm(zk)
        except Exception as e:                             # noqa: BLE001
            if world == 1:
                raise
            result["msm"] = {"error": repr(e)}
    if rank == 0 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(zk, field)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


def sumcheck_round_exchange(zk, table, out, r, world, local_rank, rehearse=False):
    """The path's real exchange step: one fused sumcheck round (fold + half sums) per rank plus ONE
    all-gather of the 2 partial sums over RCCL and the host reduction mod p.  Reported beside the fold metric."""
    import time
    import numpy as np
    import torch
    import torch.distributed as dist
    S = zk.sharded
    comm = S.Comm(device=None if rehearse else torch.device("cuda", local_rank))
    reps = 50
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    from zkmle_amd import _lib
    sums = np.zeros((2, 4), np.uint64)
    for _ in range(reps):
        _lib.check(zk.lib().zk_mle_fold_half_sums(table._h, _lib.p64(r), out._h, _lib.p64(sums), None))   # resident buffers
        g = comm.all_gather(sums)
        tot = np.stack([S.fe_sum(table.field, g[:, 0]), S.fe_sum(table.field, g[:, 1])])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return {"what": "fused sumcheck round on this rank's shard + all-gather of 2 field elements per rank (RCCL) + host reduce",
            "ms_per_round": dt * 1e3, "bytes_per_rank_per_round": 64, "collective": "all_gather",
            "backend": "gloo (rehearsal)" if rehearse else "nccl (RCCL)"}


def sharded_sumcheck_leg(zk, rank, world, local_rank, rehearse=False, log_local=20):
    """A whole GKR sumcheck (4 tables, degree 2) sharded over the ranks with device-resident rounds (include/zkmle.h zk_rounds):
    per local round one fused kernel, ONE all-reduce of 27 int64 words over RCCL and the transcript step on every rank's
    GPU; the host synchronises at the final gather and at the end only."""
    import time
    import numpy as np
    import torch
    import torch.distributed as dist
    S = zk.sharded
    comm = S.Comm(device=None if rehearse else torch.device("cuda", local_rank))
    n = 1 << log_local
    MP = zk.MultilinearPolynomial
    tabs = [[MP.random(0, n, 0x5EED0400 + 16 * rank + 2 * p + f) for f in range(2)] for p in range(2)]
    shard = S.GpuSumShard(0, tabs)
    claimed = np.zeros(4, np.uint64)
    S.sumcheck_gkr_prove_device(comm, shard, claimed, zk.Transcript())      # warm-up
    reps = 3
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        co, ch, fin = S.sumcheck_gkr_prove_device(comm, shard, claimed, zk.Transcript())
    torch.cuda.synchronize()
    dist.barrier()
    dt = (time.perf_counter() - t0) / reps
    rounds = int(co.shape[0])
    return {"what": f"GKR sumcheck on 4 tables of 2^{log_local} entries per rank ({rounds} rounds over {world} ranks), device-resident "
                    "transcript, one all-reduce(SUM) of 27 int64 words per local round", "ms_per_proof": dt * 1e3, "rounds": rounds,
            "ms_per_round": dt * 1e3 / rounds, "field_mul_per_s": 5.0 * 2 * n * world / dt,
            "backend": "gloo, host-staged (rehearsal)" if rehearse else "nccl (RCCL), on-device"}


def msm_leg(zk, args, rank, world, local_rank):
    """G1-add/s on the 2^log-n Pippenger MSM (BASELINE.json's second metric): one slice per rank, one
    all-gather of `world` affine points, world-1 additions (no bandwidth-sized collective)."""
    import time
    import numpy as np
    import torch
    import torch.distributed as dist
    n = 1 << args.log_n
    a = zk.from_ints(0, [0x5EED0003 + rank])[0]
    d = zk.from_ints(0, [0x9E3779B97F4A7C15])[0]
    bases = zk.G1Bases.synthetic(n, a, d)                   # P_i = [a + i d] G, generated on the device
    scalars = zk.MultilinearPolynomial.random(0, n, 0x5EED0003 + 97 * rank)
    out, st = zk.kzg.msm(scalars, bases, 0, True)           # warm-up
    S = zk.sharded
    comm = S.Comm(device=None if args.rehearse else torch.device("cuda", local_rank)) if world > 1 else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.msm_reps):
        out, st = zk.kzg.msm(scalars, bases, 0, True)
        stats.append(st)
        if world > 1:
            total = S.g1_sum(comm.all_gather(out))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / args.msm_reps
    if world > 1:
        tt = torch.tensor([dt], device="cpu" if args.rehearse else "cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    st = stats[-1]
    W = st["windows"]
    return {"metric": f"G1-add/s (2^{args.log_n} MSM)", "value": W * n * world / dt, "unit": "G1-add/s",
            "terms_per_s": n * world / dt, "ms_per_msm": dt * 1e3, "window_bits": st["window_bits"], "windows": W,
            "adds_per_term": W, "device_ms": {k: st[k] for k in ("ms_digits", "ms_sort", "ms_buckets", "ms_reduce", "ms_total")},
            "bound": "integer VALU (v_mad_u64_u32), not HBM", "unique_bytes_per_term": 128,
            "hbm_GBps_unique": 128.0 * n / (st["ms_total"] * 1e-3) / 1e9,
            "workload": f"2^{args.log_n} random Fr scalars x synthetic affine bases [a + i d]G per GPU; slice-sharded, all-gather of {world} points"}


def cpu_baseline_msm(zk):
    """the oracle's restatement of the reference's NAIVE commit (one 255-bit double-and-add per term,
    multilinear_kzg.rs:37-42) on 2^10 terms, one host core; the reference has no MSM routine"""
    import numpy as np
    from oracle import oracle as O
    from zkmle_amd import _lib
    n = 1 << 10
    sc = np.zeros((n, 4), np.uint64)
    _lib.check(zk.lib().zk_host_fill_random(0, 0x5EED0003, 0, n, _lib.p64(sc)))
    a = zk.from_ints(0, [0x5EED0003])[0]
    d = zk.from_ints(0, [0x9E3779B97F4A7C15])[0]
    pts = zk.G1Bases.synthetic(n, a, d).points()
    secs = O.bench_commit_naive(sc, pts)
    return {"value": n / secs, "unit": "terms/s", "cores": 1, "kind": "port",
            "sample": f"naive double-and-add commit of 2^10 terms (same generator), 1 thread; linear in the number of terms"}


def cpu_baseline(zk, field):
    """the oracle's reference-faithful single-thread fold, timed on this host on a bounded sample"""
    import numpy as np
    from oracle import oracle as O
    from zkmle_amd import _lib
    log_n = 20
    n = 1 << log_n
    tab = np.zeros((n, 4), np.uint64)
    _lib.check(zk.lib().zk_host_fill_random(field, 0x5EED0005, 0, n, _lib.p64(tab)))
    r = tab[3].copy()
    O.bench_fold(field, tab, r, 1)
    reps = 300                                  # ~10 s of single-core work
    secs = O.bench_fold(field, tab, r, reps)
    mt_reps = 100
    O.bench_fold_mt(field, tab, r, 5)
    mt_secs, threads = O.bench_fold_mt(field, tab, r, mt_reps)
    base = {"value": (n // 2) * reps / secs, "unit": "field-mul/s", "cores": 1, "kind": "port",
            "sample": f"{reps} folds of a 2^{log_n}-entry Fr table (same generator), reference allocation pattern, 1 thread"}
    mt_value = (n // 2) * mt_reps / mt_secs
    if mt_value > base["value"]:
        base["all_cores"] = {"value": mt_value, "unit": "field-mul/s", "cores": threads, "kind": "port",
                             "sample": f"{mt_reps} folds of the same table, OpenMP over output indices (the reference itself is single-threaded)"}
    else:
        base["all_cores"] = {"value": None, "cores": threads,
                             "note": "OpenMP run was slower than one thread under this host's CPU quota; not reported"}
    return base


if __name__ == "__main__":
    main()
