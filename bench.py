"""bench.py -- field-mul/s of the 2^24 MLE fold + G1-add/s of the 2^24 MSM (BASELINE.json metric) on N MI355X GPUs.

A step = one `partial_evaluate(table, 0, r)` pass (evaluation_form.rs:61-106) over one resident BLS12-381 Fr table per GPU
(2^23 field multiplications per 2^24-entry table, 96 algorithmic bytes each).

  --scaling weak   (default) every rank folds its own 2^24-entry low-bit shard; no data-path collective.
  --scaling strong BASELINE config 5: ONE 2^24-entry table low-bit-sharded N ways (2^24 / N entries per rank).
Either way the line also carries, under "config5_strong", config 5 verbatim at this N: the sharded sumcheck prover of the 2^24
table (Prover::prove rounds, prover.rs:46-63; one RCCL all-reduce per large round, whole-table absorb off) and the 2^24-term
MSM slice-sharded N ways (multilinear_kzg.rs:37-42; one all-gather of N points), and under "msm" the per-GPU 2^24 MSM.

`python bench.py --gpus N` with WORLD_SIZE unset starts the N ranks itself (child processes, one per GPU, before anything
touches a GPU in this process); under torchrun (WORLD_SIZE set) it is one rank and fails unless WORLD_SIZE == --gpus.
Prints ONE JSON line on rank 0.  See DESIGN.md section 5 for the roofline accounting.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED_TABLE, SEED_MSM = 0x5EED0005, 0x5EED0003
MSM_D = 0x9E3779B97F4A7C15


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-msm", action="store_true", help="skip the per-GPU 2^log-n MSM leg")
    ap.add_argument("--no-config5", action="store_true", help="skip the strong-scaling config-5 legs")
    ap.add_argument("--msm-reps", type=int, default=3)
    ap.add_argument("--require-rccl", action="store_true", help="exit non-zero unless the collectives ran over RCCL")
    ap.add_argument("--rehearse", action="store_true",
                    help="debug: run the N>1 code path with every rank on cuda:0 over gloo (one-GPU boxes)")
    return ap.parse_args()


def launch_ranks(args):
    """--gpus N with no WORLD_SIZE: this process becomes the launcher.  It never touches a GPU (no torch import, no HIP call):
    it starts N fresh interpreters of this file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and relays rank 0's line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + 3300
    while procs:
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in procs:                                     # a dead rank leaves the others in a collective: stop them
                    q.terminate()
        if time.time() > deadline:
            for q in procs:
                q.kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    return rc


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))

    if os.environ.get("ZK_BENCH_WATCHDOG"):                   # diagnostics: dump every thread's stack and exit if a rank is stuck
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["ZK_BENCH_WATCHDOG"]), exit=True)
    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__ as G
    zk = G.import_package()
    from zkmle_amd import _lib

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a line for the wrong rank count")
    if world & (world - 1):
        raise SystemExit("bench.py: the table shards by low index bits, --gpus must be a power of two")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: zkmle_amd has no CPU fallback")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    _lib.check(zk.lib().zk_init(local_rank))
    collectives = {"backend": "none (one rank)", "library": None, "note": None}
    if world > 1:
        if args.rehearse:
            dist.init_process_group("gloo")
            collectives = {"backend": "gloo (rehearsal: every rank on cuda:0)", "library": None, "note": "not an RCCL measurement"}
        else:
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                probe = torch.zeros(1, device="cuda")
                dist.all_reduce(probe)                      # the first collective builds the communicator: fail here, not mid-run
                torch.cuda.synchronize()
                collectives = {"backend": "nccl (RCCL over xGMI)", "library": None, "note": None}
            except Exception as e:                          # noqa: BLE001
                if args.require_rccl:
                    raise SystemExit(f"bench.py --require-rccl: nccl (RCCL) process group failed: {e!r}")
                note = f"nccl (RCCL) unavailable, collectives over gloo with host staging: {e!r}"[:300]
                try:
                    dist.destroy_process_group()
                except Exception:                           # noqa: BLE001
                    pass
                os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
                dist.init_process_group("gloo")
                args.rehearse = True                        # host-staged tensors from here on (each rank keeps its own GPU)
                collectives = {"backend": "gloo (FALLBACK, host staged)", "library": None, "note": note}
    S = zk.sharded
    comm = S.Comm(device=None if (args.rehearse or world == 1) else torch.device("cuda", local_rank))
    if world > 1:
        collectives["library"] = comm.native_backend()      # "rccl": the library's own ncclCommInitRank communicator
        if comm.native_note:                                # it could not be created: the provers' exchange goes over gloo, loudly
            collectives["note"] = comm.native_note
            collectives["backend"] += " + gloo (FALLBACK for the provers' exchange, host staged)"
        if args.require_rccl and collectives["library"] != "rccl":
            raise SystemExit("bench.py --require-rccl: the provers' communicator is not RCCL")

    field = zk.FR381
    n_global = 1 << args.log_n
    strong = args.scaling == "strong"
    n = n_global // world if strong else n_global          # this rank's table
    if n < 4:
        raise SystemExit("table too small for this many ranks")
    half = n // 2
    MP = zk.MultilinearPolynomial
    L = zk.lib()
    table = MP.alloc(field, n)                               # shard-wise on-device generation
    if strong:
        first, stride, seed = rank, world, SEED_TABLE        # local j <- global j * world + rank of ONE table
    else:
        first, stride, seed = 0, 1, SEED_TABLE + rank        # an own table per rank
    _lib.check(L.zk_table_fill_random_strided(table._h, seed, first, stride))
    out = MP.alloc(field, half)
    r = np.zeros(4, np.uint64)
    _lib.check(L.zk_host_fill_random(field, SEED_TABLE, n_global, 1, _lib.p64(r)))
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        _lib.check(L.zk_mle_fold(table._h, 0, _lib.p64(r), out._h, stream))

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
    # HIP events on the launch stream, one every `chunk` launches: mean over the timed region + median of >= 20 chunk means
    chunk = max(1, args.steps // 25)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range((args.steps + chunk - 1) // chunk + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        step()
        if (k + 1) % chunk == 0 or k + 1 == args.steps:
            marks[(k + chunk) // chunk].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms = marks[0].elapsed_time(marks[-1]) / args.steps
    per_chunk = []
    for i in range(len(marks) - 1):
        cnt = min(chunk, args.steps - i * chunk)
        per_chunk.append(marks[i].elapsed_time(marks[i + 1]) / cnt)
    kern_ms_median = statistics.median(per_chunk)
    if world > 1:
        tt = torch.tensor([dt], device="cpu" if dist.get_backend() == "gloo" else "cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    muls = half * args.steps * world
    value = muls / dt
    algo_bytes = 96.0 * half                               # per launch: 2 x 32 B read + 32 B write per mul
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
    per = f"2^{args.log_n}-entry table per GPU" if not strong else f"ONE 2^{args.log_n}-entry table, 2^{args.log_n}/{world} entries per GPU"
    result = {
        "metric": f"field-mul/s (2^{args.log_n} MLE fold)", "value": value, "unit": "field-mul/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": f"{args.log_n}-variable MLE fold (partial_evaluate var 0), BLS12-381 Fr, {per}",
                   "log_n": args.log_n, "field": "bls12_381_fr",
                   "arithmetic": "255-bit Montgomery field, 8 x u32 limbs in HBM, products as 29-bit-limb v_mad_u64_u32 scans",
                   "sharding": "low-bit shard per rank, no data-path collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                     "traffic": None, "kernel": "fold0_kernel<Fr381>", "kernel_ms": kern_ms, "kernel_ms_median": kern_ms_median,
                     "achieved_median": algo_bytes / (kern_ms_median * 1e-3) / 1e9, "timing_chunks": len(per_chunk),
                     "algorithmic_bytes_per_launch": algo_bytes},
        "collectives": collectives,
    }
    for rnd in ("r2", "r1"):                                 # PMC passes are separate rocprofv3 runs (committed summary)
        pmc = os.path.join(ROOT, "profiles", rnd, "fold_2p24_pmc.json")
        if args.log_n == 24 and not strong and os.path.exists(pmc):
            with open(pmc) as f:
                result["roofline"]["traffic"] = json.load(f)["hbm_bytes_per_launch"]
            result["roofline"]["traffic_source"] = (f"profiles/{rnd}/fold_2p24_pmc.json (committed summary of separate rocprofv3 --pmc FETCH_SIZE / "
                                                    "WRITE_SIZE passes, FETCH doubled per the gfx950 note; not re-measured in this run)")
            break
    if world > 1:
        try:                                               # a failure in a secondary leg must not cost the headline line
            result["sharded_sumcheck"] = sharded_gkr_sumcheck_leg(zk, comm, rank, world, collectives)
        except Exception as e:                             # noqa: BLE001
            result["sharded_sumcheck"] = {"error": repr(e)}
    if not args.no_config5:
        try:
            result["config5_strong"] = config5_leg(zk, comm, args, rank, world, collectives)
        except Exception as e:                             # noqa: BLE001
            if world == 1:
                raise
            result["config5_strong"] = {"error": repr(e)}
    if not args.no_msm:
        try:
            result["msm"] = msm_leg(zk, comm, args, rank, world)
            if rank == 0 and not args.no_cpu_baseline:
                result["msm"]["cpu_baseline"] = cpu_baseline_msm(zk)
        except Exception as e:                             # noqa: BLE001
            if world == 1:
                raise
            result["msm"] = {"error": repr(e)}
    if rank == 0 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(zk, field)
        # what was timed is a correct fold: sampled outputs of the last launch against the oracle on the host mirror of the inputs
        result["post_check"] = post_check_fold(zk, field, out, r, seed, first, stride, half)
    if rank == 0:
        print(json.dumps(result), flush=True)
    comm.close()
    if world > 1:
        dist.destroy_process_group()


def _barrier_time(world, fn, reps, rehearse):
    """max over ranks of the mean wall time of `fn` over `reps` calls between barriers"""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = [fn() for _ in range(reps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / reps
    if world > 1:                                          # the default group's backend decides where its tensors live
        tt = torch.tensor([dt], device="cpu" if dist.get_backend() == "gloo" else "cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, outs[-1]


def config5_leg(zk, comm, args, rank, world, collectives):
    """BASELINE config 5 at this rank count (strong scaling): the 2^log-n table low-bit-sharded over the ranks through the
    sharded sumcheck prover (include/zkmle.h zk_sharded_sumcheck_basic_prove, whole-table absorb off) and the 2^log-n-term MSM
    slice-sharded (zk_sharded_msm_g1)."""
    import numpy as np
    from zkmle_amd import _lib
    S = zk.sharded
    Lb = zk.lib()
    field = 0
    n_global = 1 << args.log_n
    n = n_global // world
    out = {"what": f"one 2^{args.log_n}-entry Fr table and one 2^{args.log_n}-term MSM sharded {world}-way", "backend": collectives["backend"],
           "library_comm": collectives["library"]}
    # -- sumcheck prover rounds on the sharded table
    table = zk.MultilinearPolynomial.alloc(field, n)
    _lib.check(Lb.zk_table_fill_random_strided(table._h, SEED_TABLE, rank, world))
    shard = S.GpuShard(table)
    S.sumcheck_basic_prove_device(comm, shard, absorb_table=False)            # warm-up
    rx0, nc0 = comm.native_stats()
    dt, (claimed, rp, ch) = _barrier_time(world, lambda: S.sumcheck_basic_prove_device(comm, shard, absorb_table=False), 5, args.rehearse)
    rx1, nc1 = comm.native_stats()
    # verifier equations (verifier.rs:47-70) on the proof just timed: claim chain + the table evaluated at the challenges
    cur, ok = claimed, True
    for k in range(rp.shape[0]):
        ok = ok and np.array_equal(S.fe_add(field, rp[k, 0], rp[k, 1]), cur)
        d = np.zeros(4, np.uint64)
        _lib.check(Lb.zk_fe_sub(field, _lib.p64(rp[k, 1]), _lib.p64(rp[k, 0]), _lib.p64(d)))
        m = np.zeros(4, np.uint64)
        _lib.check(Lb.zk_fe_mul(field, _lib.p64(np.ascontiguousarray(ch[k])), _lib.p64(d), _lib.p64(m)))
        cur = S.fe_add(field, rp[k, 0], m)
    ok = ok and np.array_equal(S.mle_evaluate(comm, table, ch), cur)
    out["sumcheck"] = {"what": f"Prover::prove rounds of the 2^{args.log_n} table ({rp.shape[0]} rounds), 2^{args.log_n}/{world} entries per rank, "
                               "up to 4 rounds per pass over the shard: one all-reduce(SUM) of 2^m x 9 int64 words (m <= 4 rounds' segment sums) on the prover's stream per pass, replicated one-launch tail, transcript steps on each rank's host through the mailbox",
                       "ms_per_proof": dt * 1e3, "field_mul_per_s": (n_global - 1) / dt, "rounds": int(rp.shape[0]),
                       "collectives_per_proof": (nc1 - nc0) // 5, "bytes_received_per_proof": (rx1 - rx0) // 5,
                       "verifier_equations_hold": bool(ok)}
    del shard, table
    # -- MSM, terms sliced over the ranks: rank g owns the terms [g n / G, (g + 1) n / G)
    lo = rank * n
    a = zk.from_ints(0, [SEED_MSM])[0]
    d = zk.from_ints(0, [MSM_D])[0]
    a_lo = np.zeros(4, np.uint64)
    lo_fe = zk.from_ints(0, [lo])[0]
    _lib.check(Lb.zk_fe_mul(0, _lib.p64(lo_fe), _lib.p64(d), _lib.p64(a_lo)))
    a_lo = S.fe_add(0, a, a_lo)                                               # P_i = [a + i d] G for the global index i
    bases = zk.G1Bases.synthetic(n, a_lo, d)
    scalars = zk.MultilinearPolynomial.alloc(0, n)
    _lib.check(Lb.zk_table_fill_random_strided(scalars._h, SEED_MSM, lo, 1))
    S.msm_device(comm, scalars, bases, 0, True)                               # warm-up
    dt, (pt, st) = _barrier_time(world, lambda: S.msm_device(comm, scalars, bases, 0, True), args.msm_reps, args.rehearse)
    same = np.array_equal(comm.all_gather(pt), np.broadcast_to(pt, (world, 12)))
    W = st["windows"]
    out["msm"] = {"what": f"2^{args.log_n}-term MSM, 2^{args.log_n}/{world} terms per rank, one all-gather of {world} affine points + {world - 1} additions",
                  "ms_per_msm": dt * 1e3, "g1_add_per_s": W * n_global / dt, "terms_per_s": n_global / dt, "window_bits": st["window_bits"],
                  "windows": W, "local_device_ms": st["ms_total"], "same_point_on_every_rank": bool(same),
                  "point_x_limb0": int(pt[0])}
    return out


def sharded_gkr_sumcheck_leg(zk, comm, rank, world, collectives, log_local=20):
    """A whole GKR sumcheck (4 tables, degree 2; sumcheck_gkr_protocol.rs:24-67) sharded over the ranks, weak: 2^20 entries per
    table per rank.  One C-ABI call per proof (zk_sharded_sumcheck_gkr_prove)."""
    import numpy as np
    S = zk.sharded
    n = 1 << log_local
    MP = zk.MultilinearPolynomial
    tabs = [[MP.random(0, n, 0x5EED0400 + 16 * rank + 2 * p + f) for f in range(2)] for p in range(2)]
    shard = S.GpuSumShard(0, tabs)
    claimed = np.zeros(4, np.uint64)
    S.sumcheck_gkr_prove_device(comm, shard, claimed, zk.Transcript())      # warm-up
    dt, (co, ch, fin) = _barrier_time(world, lambda: S.sumcheck_gkr_prove_device(comm, shard, claimed, zk.Transcript()), 3,
                                      collectives["backend"].startswith("gloo"))
    rounds = int(co.shape[0])
    return {"what": f"GKR sumcheck on 4 tables of 2^{log_local} entries per rank ({rounds} rounds over {world} ranks), "
                    "one all-reduce(SUM) of 27 int64 words per large round", "ms_per_proof": dt * 1e3, "rounds": rounds,
            "ms_per_round": dt * 1e3 / rounds, "field_mul_per_s": 5.0 * 2 * n * world / dt, "backend": collectives["backend"]}


def msm_leg(zk, comm, args, rank, world):
    """G1-add/s on the 2^log-n Pippenger MSM (BASELINE.json's second metric), weak: one 2^log-n-term MSM per rank, then one
    all-gather of `world` affine points + world-1 additions (no bandwidth-sized collective)."""
    import numpy as np
    n = 1 << args.log_n
    a = zk.from_ints(0, [SEED_MSM + rank])[0]
    d = zk.from_ints(0, [MSM_D])[0]
    bases = zk.G1Bases.synthetic(n, a, d)                   # P_i = [a + i d] G, generated on the device
    scalars = zk.MultilinearPolynomial.random(0, n, SEED_MSM + 97 * rank)
    S = zk.sharded
    S.msm_device(comm, scalars, bases, 0, True)             # warm-up
    dt, (pt, st) = _barrier_time(world, lambda: S.msm_device(comm, scalars, bases, 0, True), args.msm_reps, args.rehearse)
    W = st["windows"]
    res = {"metric": f"G1-add/s (2^{args.log_n} MSM)", "value": W * n * world / dt, "unit": "G1-add/s",
           "terms_per_s": n * world / dt, "ms_per_msm": dt * 1e3, "window_bits": st["window_bits"], "windows": W,
           "adds_per_term": W, "device_ms": {k: st[k] for k in ("ms_digits", "ms_sort", "ms_buckets", "ms_reduce", "ms_total")},
           "unique_bytes_per_term": 128, "hbm_GBps_unique": 128.0 * n / (st["ms_total"] * 1e-3) / 1e9,
           "workload": f"2^{args.log_n} random Fr scalars x synthetic affine bases [a + i d]G per GPU; one Pippenger per rank, all-gather of {world} points"}
    try:
        res["roofline"] = msm_roofline(st)
    except Exception as e:                                  # noqa: BLE001
        res["roofline"] = {"error": repr(e)}
    return res


# v_mad_u64_u32 per lane of one mixed addition of msm_bucket_sum_kernel (csrc/g1u.cuh g1u_madd on 14 x 29-bit limbs of Fq381):
# 6 products (392) + 2 squarings (301) + 1 dual product with one reduction (588).  Source-level count = the ISA's common path
# (tools/count_mads.py, profiles/r2/msm_mad_count.txt).
MADS_PER_MIXED_ADD = 6 * 392 + 2 * 301 + 588


def msm_roofline(st):
    """integer-VALU roofline of the dominant MSM kernel (msm_bucket_sum_kernel): achieved v_mad_u64_u32 lane-ops/s from the
    bucket phase's HIP-event time and the counted multiply-adds per mixed addition, against the rate the chip sustains on a
    register-resident v_mad_u64_u32 chain, re-measured in this run (tools/microbench, `mad_u64_u32`)."""
    adds = float(st["entries"])                             # one mixed addition per (term, window) entry with a non-zero digit
    achieved = adds * MADS_PER_MIXED_ADD / (st["ms_buckets"] * 1e-3)
    peak, src = measured_mad_peak()
    return {"bound": "valu v_mad_u64_u32", "kernel": "msm_bucket_sum_kernel", "achieved": achieved / 1e12, "peak": peak / 1e12, "unit": "Tmad/s",
            "frac": achieved / peak, "mixed_adds": adds, "mads_per_mixed_add": MADS_PER_MIXED_ADD, "kernel_ms": st["ms_buckets"],
            "peak_source": src, "traffic": None}


def measured_mad_peak():
    exe = os.path.join(ROOT, "tools", "microbench")
    if os.path.exists(exe):
        try:
            txt = subprocess.run([exe, "--only-mad"], capture_output=True, text=True, timeout=120).stdout
            rates = [float(json.loads(line)["lane_ops_per_s"]) for line in txt.splitlines()
                     if line.startswith("{") and '"k_mad_u64_u32"' in line]
            if rates:
                return max(rates), "tools/microbench --only-mad (register-resident v_mad_u64_u32 chains on every CU), measured in this run"
        except Exception:                                   # noqa: BLE001
            pass
    return 3.13e13, "profiles/r1/microbench_instr_rates.jsonl (committed; the in-run micro-benchmark was unavailable)"


def cpu_baseline_msm(zk):
    """the oracle's restatement of the reference's NAIVE commit (one 255-bit double-and-add per term,
    multilinear_kzg.rs:37-42) on 2^10 terms, one host core; the reference has no MSM routine.  Beside it, a CPU Pippenger
    (same bucket method as the GPU path, OpenMP over windows) on all host cores: BASELINE.md section 3.2."""
    import numpy as np
    from oracle import oracle as O
    from zkmle_amd import _lib
    n = 1 << 10
    sc = np.zeros((n, 4), np.uint64)
    _lib.check(zk.lib().zk_host_fill_random(0, SEED_MSM, 0, n, _lib.p64(sc)))
    a = zk.from_ints(0, [SEED_MSM])[0]
    d = zk.from_ints(0, [MSM_D])[0]
    pts = zk.G1Bases.synthetic(n, a, d).points()
    secs = O.bench_commit_naive(sc, pts)
    base = {"value": n / secs, "unit": "terms/s", "cores": 1, "kind": "port",
            "sample": "naive double-and-add commit of 2^10 terms (same generator), 1 thread; linear in the number of terms"}
    if hasattr(O, "bench_pippenger_mt"):
        m = 1 << 18                                         # ~10 s of CPU work over all cores
        sc = np.zeros((m, 4), np.uint64)
        _lib.check(zk.lib().zk_host_fill_random(0, SEED_MSM, 0, m, _lib.p64(sc)))
        pts = zk.G1Bases.synthetic(m, a, d).points()
        secs, threads, c = O.bench_pippenger_mt(sc, pts)
        base["all_cores"] = {"value": m / secs, "unit": "terms/s", "cores": threads, "kind": "port", "window_bits": c,
                             "g1_add_per_s": m * ((255 + c - 1) // c) / secs,
                             "sample": f"CPU Pippenger (signed {c}-bit windows, OpenMP over windows) of 2^18 terms (same generator)"}
    return base


def cpu_baseline(zk, field):
    """the oracle's reference-faithful single-thread fold, timed on this host on a bounded sample"""
    import numpy as np
    from oracle import oracle as O
    from zkmle_amd import _lib
    log_n = 20
    n = 1 << log_n
    tab = np.zeros((n, 4), np.uint64)
    _lib.check(zk.lib().zk_host_fill_random(field, SEED_TABLE, 0, n, _lib.p64(tab)))
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


def post_check_fold(zk, field, out, r, seed, first, stride, half, samples=512):
    """the oracle (checker) folds the host mirror of `samples` input pairs; the timed kernel's output must equal it bit for bit"""
    import numpy as np
    from oracle import oracle as O
    from zkmle_amd import _lib
    rng = np.random.default_rng(1)
    samples = min(samples, half)                            # a power of two (the oracle's table must be one): one index per stratum
    width = half // samples
    idx = np.arange(samples) * width + rng.integers(0, width, samples)
    idx[0], idx[-1] = 0, half - 1
    pairs = np.zeros((2 * len(idx), 4), np.uint64)
    one = np.zeros((1, 4), np.uint64)
    for k, j in enumerate(idx):
        for h, jj in ((0, int(j)), (1, int(j) + half)):
            _lib.check(zk.lib().zk_host_fill_random(field, seed, first + jj * stride, 1, _lib.p64(one)))
            pairs[k + h * len(idx)] = one[0]
    want = O.partial_evaluate(field, pairs, 0, r)           # pairs entry k with entry k + len(idx)
    got = out.evaluated_values[idx]
    ok = bool(np.array_equal(got, want))
    if not ok:
        raise SystemExit("bench.py: the timed fold's output differs from the oracle on the sampled entries")
    return {"checked_entries": int(len(idx)), "bit_exact_vs_oracle": ok}


if __name__ == "__main__":
    main()
