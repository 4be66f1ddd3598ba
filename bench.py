"""bench.py -- field-mul/s of the 2^24 MLE fold + G1-add/s of the 2^24 MSM (BASELINE.json metric) on N MI355X GPUs.

A step = one `partial_evaluate(table, 0, r)` pass (evaluation_form.rs:61-106) over a resident BLS12-381 Fr table
(one field multiplication and 96 algorithmic bytes per output entry).

  N = 1   the 2^24-entry table on the one GPU (2^23 field-mul per step).
  N > 1   BASELINE config 5, strong scaling (the default): ONE 2^24-entry table low-bit-sharded N ways, every rank
          folds its 2^24 / N entries per step -- `value` = 2^23 x K / time, with the whole sharded sumcheck proof of that table
          (Prover::prove rounds, prover.rs:46-63; one all-reduce per pass) and the 2^24-term MSM slice-sharded N ways
          (multilinear_kzg.rs:37-42; one all-gather of N points) beside it under "config5_strong".  `--scaling weak` makes the
          headline a 2^24 table per GPU instead; by default that run is the secondary key "weak".
The line also carries "msm" (one 2^24-term MSM per GPU) and, at N = 1, "configs": BASELINE configs 2, 3 and 4 timed and checked.
Everything that is timed is checked afterwards: the fold against the oracle on sampled entries, every proof through the
verifier's equations, every MSM through the O(N) identity of its structured bases.

`python bench.py --gpus N` with WORLD_SIZE unset starts the N ranks itself (child processes, one per GPU, before anything
touches a GPU in this process); under torchrun (WORLD_SIZE set) it is one rank and fails unless WORLD_SIZE == --gpus.
`--rehearse` runs the N-rank code path on ONE GPU with the ranks as threads of this process (a one-GPU box allows at most 6
processes on its card, config 5 has 8 ranks), collectives over the library's ranks-as-threads transport.
Prints ONE JSON line on rank 0.  See DESIGN.md section 5 for the roofline accounting.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED_TABLE, SEED_MSM = 0x5EED0005, 0x5EED0003
MSM_D = 0x9E3779B97F4A7C15
HBM_PEAK_GBPS = 8000.0                                       # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="N > 1 only; default strong = BASELINE config 5 (ONE table sharded N ways)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-msm", action="store_true", help="skip the per-GPU 2^log-n MSM leg")
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 legs (sharded proof + sharded MSM)")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs 2, 3, 4 (N = 1 only)")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the secondary weak-scaling fold")
    ap.add_argument("--msm-reps", type=int, default=3)
    ap.add_argument("--require-rccl", action="store_true",
                    help="exit non-zero unless the collectives ran over RCCL and every config-5 leg succeeded")
    ap.add_argument("--preflight", action="store_true",
                    help="N > 1: build the communicators, run every collective the provers use once on known data and one small sharded proof / MSM "
                         "of each kind against the checker, print one JSON line and exit (non-zero on any mismatch): seconds, before a long run")
    ap.add_argument("--rehearse", action="store_true",
                    help="debug: the N > 1 code path with every rank a thread of this process on cuda:0 (one-GPU boxes)")
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


class RankEnv:
    """One rank's view of the job: who it is, how it meets the other ranks (torch.distributed between processes, a thread
    barrier between the threads of a rehearsal), and its end of the provers' communicator."""

    def __init__(self, rank, world, comm, collectives, dist=None, tgroup=None):
        self.rank, self.world, self.comm, self.collectives = rank, world, comm, collectives
        self.dist, self.tgroup = dist, tgroup

    def barrier(self):
        if self.world == 1:
            return
        if self.tgroup is not None:
            self.tgroup["barrier"].wait()
        else:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if self.world == 1:
            return float(x)
        if self.tgroup is not None:
            slots = self.tgroup["slots"]
            slots[self.rank] = float(x)
            self.tgroup["barrier"].wait()
            m = max(slots)
            self.tgroup["barrier"].wait()
            return m
        import torch
        tt = torch.tensor([x], device="cpu" if self.dist.get_backend() == "gloo" else "cuda", dtype=torch.float64)
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())

    def timed(self, fn, reps):
        """max over ranks of the mean wall time of `fn` over `reps` calls between barriers; -> (seconds, last result)"""
        import torch
        self.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = [fn() for _ in range(reps)]
        torch.cuda.synchronize()
        self.barrier()
        dt = (time.perf_counter() - t0) / reps
        return self.max_over_ranks(dt), outs[-1]


def main():
    args = parse_args()
    if args.gpus > 1 and args.gpus & (args.gpus - 1):
        raise SystemExit("bench.py: the table shards by low index bits, --gpus must be a power of two")
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={os.environ['WORLD_SIZE']} but --gpus {args.gpus}: refusing to report a line for the wrong rank count")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.rehearse:
        sys.exit(launch_ranks(args))
    if args.scaling is None:
        args.scaling = "strong" if args.gpus > 1 else "weak"

    if os.environ.get("ZK_BENCH_WATCHDOG"):                   # diagnostics: dump every thread's stack and exit if a rank is stuck
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["ZK_BENCH_WATCHDOG"]), exit=True)
    import torch
    import __graft_entry__ as G
    zk = G.import_package()
    from zkmle_amd import _lib
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: zkmle_amd has no CPU fallback")

    if args.rehearse and args.gpus > 1:
        return rehearse_threads(args, zk)

    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    _lib.check(zk.lib().zk_init(local_rank))
    collectives = {"backend": "none (one rank)", "library": None, "note": None}
    device = None
    if world > 1:
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            probe = torch.zeros(1, device="cuda")
            dist.all_reduce(probe)                      # the first collective builds the communicator: fail here, not mid-run
            torch.cuda.synchronize()
            collectives = {"backend": "nccl (RCCL over xGMI)", "library": None, "note": None}
            device = torch.device("cuda", local_rank)
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
            collectives = {"backend": "gloo (FALLBACK, host staged)", "library": None, "note": note}
    comm = zk.sharded.Comm(device=device)
    if world > 1:
        collectives["library"] = comm.native_backend()      # "rccl": the library's own ncclCommInitRank communicator
        if comm.native_note:                                # it could not be created: the provers' exchange goes over gloo, loudly
            collectives["note"] = comm.native_note
            collectives["backend"] += " + gloo (FALLBACK for the provers' exchange, host staged)"
        if args.require_rccl and collectives["library"] != "rccl":
            raise SystemExit("bench.py --require-rccl: the provers' communicator is not RCCL")
    env = RankEnv(rank, world, comm, collectives, dist=dist)
    if args.preflight:
        line = preflight(env, zk)
        if rank == 0:
            print(json.dumps(line), flush=True)
        comm.close()
        if world > 1:
            dist.destroy_process_group()
        sys.exit(0 if line["ok"] else 4)
    result = run_rank(env, args, zk)
    if rank == 0:
        print(json.dumps(result), flush=True)
    comm.close()
    if world > 1:
        dist.destroy_process_group()
    if result.get("failed_legs") and args.require_rccl:
        raise SystemExit("bench.py --require-rccl: a config-5 leg failed: " + ", ".join(result["failed_legs"]))


def rehearse_threads(args, zk):
    """The N-rank code path on one GPU: N threads of this process, each with its own HIP stream, exchanging through the
    library's ranks-as-threads transport (include/zkmle.h zk_comm_local_group_*).  Not an RCCL measurement; says so."""
    import ctypes as C
    import torch
    from zkmle_amd import _lib
    world = args.gpus
    group = zk.sharded.LocalGroup(world)
    tgroup = {"barrier": threading.Barrier(world), "slots": [0.0] * world}
    results, errors = [None] * world, []
    collectives = {"backend": f"local-threads (rehearsal: {world} ranks as threads on cuda:0, host-staged exchanges)", "library": None,
                   "note": "not an RCCL measurement: one GPU shared by every rank"}

    def rank_main(rank):
        comm = None
        try:
            torch.cuda.set_device(0)
            lib = zk.lib()
            _lib.check(lib.zk_init(0))
            st = torch.cuda.Stream()
            torch.cuda.set_stream(st)
            lib.zk_set_stream.argtypes = [C.c_void_p]
            _lib.check(lib.zk_set_stream(C.c_void_p(st.cuda_stream)))
            comm = group.comm(rank)
            coll = dict(collectives, library=comm.native_backend())
            renv = RankEnv(rank, world, comm, coll, tgroup=tgroup)
            results[rank] = preflight(renv, zk) if args.preflight else run_rank(renv, args, zk)
        except BaseException as e:                        # noqa: BLE001
            import traceback
            errors.append((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
            group.abort()
            tgroup["barrier"].abort()
        finally:
            try:
                torch.cuda.synchronize()
                zk.lib().zk_set_stream(None)
                if comm is not None:
                    comm.close()
            except Exception:                             # noqa: BLE001
                pass

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    group.close()
    if errors:
        for rank, tb in errors:
            print(f"rank {rank} failed:\n{tb}", file=sys.stderr)
        raise SystemExit(1)
    print(json.dumps(results[0]), flush=True)
    if args.preflight and not results[0]["ok"]:
        raise SystemExit(4)


# ---- N > 1 preflight ----------------------------------------------------------------------------------------------------------------
def preflight(env, zk):
    """Everything a multi-rank run depends on, in seconds: the provers' communicator (RCCL, or what stands in for it) carries each collective
    the provers use -- the int64 limb all-reduce (exactness past 2^32), the all-gather (rank order), the broadcast of the 208-byte sponge
    (from the first and from the last rank) -- on known data; then one small proof / MSM of each sharded kind against the checker: the
    basic sumcheck WITH the whole-table absorb on a table of four absorb chunks (the chunked gather's order), the GKR sumcheck, evaluate and the
    MSM.  -> {"ok": bool, "checks": {...}} on every rank (the verdict is agreed: one more exchange at the end)."""
    import ctypes as C
    import numpy as np
    import torch
    from oracle import oracle as O
    from zkmle_amd import _lib
    S = zk.sharded
    lib = S._declare_host()
    rank, world, comm = env.rank, env.world, env.comm
    h = comm.native()
    t0 = time.perf_counter()
    checks = {}
    dev = torch.device("cuda", torch.cuda.current_device())

    def dptr(t):
        return C.c_void_p(t.data_ptr())

    # 1. all-reduce(SUM, int64) of limb words: every rank contributes 2^32 - 1 - i, so the sums pass 2^32 (27 words = one GKR round; 2^7 x 9 = the largest pass)
    for count in (27, 9 << 7):
        v = torch.tensor([0xFFFFFFFF - i for i in range(count)], dtype=torch.int64, device=dev)
        _lib.check(lib.zk_comm_all_reduce_sum_i64(h, dptr(v), count))
        torch.cuda.synchronize()
        want = np.array([world * (0xFFFFFFFF - i) for i in range(count)], np.int64)
        checks[f"all_reduce_i64_{count}_words_exact"] = bool(np.array_equal(v.cpu().numpy(), want))
    # 2. all-gather: 96 bytes per rank (an affine point), slot r must hold rank r's bytes
    send = torch.full((96,), rank + 1, dtype=torch.uint8, device=dev)
    recv = torch.zeros((world * 96,), dtype=torch.uint8, device=dev)
    _lib.check(lib.zk_comm_all_gather(h, dptr(send), dptr(recv), 96))
    torch.cuda.synchronize()
    got = recv.cpu().numpy().reshape(world, 96)
    checks["all_gather_rank_order"] = bool(all((got[r] == r + 1).all() for r in range(world)))
    # 3. broadcast of 208 bytes (25 sponge lanes + fill) from the first and from the last rank
    for root in sorted({0, world - 1}):
        b = torch.tensor([(7 * i + root) % 251 if rank == root else 0 for i in range(208)], dtype=torch.uint8, device=dev)
        _lib.check(lib.zk_comm_broadcast(h, dptr(b), 208, root))
        torch.cuda.synchronize()
        checks[f"broadcast_208_from_rank_{root}"] = bool(np.array_equal(b.cpu().numpy(), np.array([(7 * i + root) % 251 for i in range(208)], np.uint8)))
    # 4. basic sumcheck with the whole-table absorb: ONE 2^20-entry table = four chunks of the streamed gather to rank 0
    logn = 20
    full = np.zeros((1 << logn, 4), np.uint64)
    _lib.check(zk.lib().zk_host_fill_random(0, 0x5EED0F00, 0, 1 << logn, _lib.p64(full)))
    shard = S.GpuShard.from_array(0, S.shard_of(full, rank, world))
    cs, rp, ch = S.sumcheck_basic_prove_device(comm, shard, absorb_table=True)
    wcs, wrp, wch = O.sumcheck_basic_prove(0, full)                            # the checker, on the whole table
    checks["sharded_basic_sumcheck_with_absorb_equals_oracle"] = bool(np.array_equal(cs, wcs) and np.array_equal(rp, wrp) and np.array_equal(ch, wch))
    point = full[:logn].copy()
    checks["sharded_evaluate_equals_oracle"] = bool(np.array_equal(S.mle_evaluate(comm, shard.poly, point), O.evaluate(0, full, point)))
    # 5. GKR sumcheck on 2 x 2 tables of 2^14 entries
    lg = 14
    tabs = np.zeros((2, 2, 1 << lg, 4), np.uint64)
    for k in range(4):
        _lib.check(zk.lib().zk_host_fill_random(0, 0x5EED0F10 + k, 0, 1 << lg, _lib.p64(tabs[k // 2, k % 2])))
    claimed = O.vec_sum(0, O.sumpoly_reduce(0, tabs))
    ss = S.GpuSumShard(0, [[zk.MultilinearPolynomial(0, S.shard_of(tabs[p, f], rank, world)) for f in range(2)] for p in range(2)])
    co, gch, _ = S.sumcheck_gkr_prove_device(comm, ss, claimed, zk.Transcript())
    wco, wgch = O.sumcheck_gkr_prove(0, tabs, claimed, O.Transcript())
    checks["sharded_gkr_sumcheck_equals_oracle"] = bool(np.array_equal(co, wco) and np.array_equal(gch, wgch))
    # 6. MSM: 2^12 terms sliced over the ranks, the identity of the structured bases on the whole
    n = 1 << 12
    per = n // world
    a_lo = zk.from_ints(0, [SEED_MSM + rank * per * MSM_D])[0]
    bases = zk.G1Bases.synthetic(per, a_lo, zk.from_ints(0, [MSM_D])[0])
    sc = zk.MultilinearPolynomial.alloc(0, per)
    _lib.check(zk.lib().zk_table_fill_random_strided(sc._h, SEED_MSM, rank * per, 1))
    pt = S.msm_device(comm, sc, bases)
    whole = zk.MultilinearPolynomial.alloc(0, n)
    _lib.check(zk.lib().zk_table_fill_random_strided(whole._h, SEED_MSM, 0, 1))
    checks["sharded_msm_identity"] = bool(msm_identity_check(zk, whole, SEED_MSM, MSM_D, pt))
    mine = all(checks.values())
    agreed = comm.all_gather(np.array([1 if mine else 0], np.uint64))
    checks["every_rank_agrees"] = bool(np.asarray(agreed).reshape(-1).all())
    return {"preflight": True, "ok": bool(all(checks.values())), "n_gpus": world, "collectives": env.collectives, "checks": checks,
            "seconds": time.perf_counter() - t0}


# ---- the fold (headline) ------------------------------------------------------------------------------------------------
def fold_leg(env, zk, n, seed, first, stride, r, steps, warmup, prewarm_s=0.4):
    """K timed launches of partial_evaluate(table, 0, r) on this rank's n-entry table between barriers.  HIP events on the launch
    stream, one every K/25 launches: mean over the timed region + median of the chunk means.  -> (numbers, table, out)"""
    import torch
    from zkmle_amd import _lib
    L = zk.lib()
    MP = zk.MultilinearPolynomial
    table = MP.alloc(zk.FR381, n)                            # shard-wise on-device generation
    _lib.check(L.zk_table_fill_random_strided(table._h, seed, first, stride))
    out = MP.alloc(zk.FR381, n // 2)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        _lib.check(L.zk_mle_fold(table._h, 0, _lib.p64(r), out._h, stream))

    # clock pre-warm (untimed, outside the W / K protocol, reported in the line as `prewarm_s`): a cold MI355X needs a few hundred
    # milliseconds of load before it holds its sustained clock; without this a short K reads 20 % low
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < prewarm_s:
        for _ in range(50):
            step()
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    env.barrier()
    torch.cuda.synchronize()
    chunk = max(1, steps // 25)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range((steps + chunk - 1) // chunk + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(steps):
        step()
        if (k + 1) % chunk == 0 or k + 1 == steps:
            marks[(k + chunk) // chunk].record()
    torch.cuda.synchronize()
    env.barrier()
    torch.cuda.synchronize()
    dt = env.max_over_ranks(time.perf_counter() - t0)
    kern_ms = marks[0].elapsed_time(marks[-1]) / steps
    per_chunk = []
    for i in range(len(marks) - 1):
        cnt = min(chunk, steps - i * chunk)
        per_chunk.append(marks[i].elapsed_time(marks[i + 1]) / cnt)
    return {"dt": dt, "kernel_ms": kern_ms, "kernel_ms_median": statistics.median(per_chunk), "chunks": len(per_chunk),
            "prewarm_s": prewarm_s}, table, out


def run_rank(env, args, zk):
    import numpy as np
    from zkmle_amd import _lib
    rank, world = env.rank, env.world
    field = zk.FR381
    n_global = 1 << args.log_n
    strong = args.scaling == "strong" and world > 1
    n = n_global // world if strong else n_global          # this rank's table
    if n < 4:
        raise SystemExit("table too small for this many ranks")
    half = n // 2
    r = np.zeros(4, np.uint64)
    _lib.check(zk.lib().zk_host_fill_random(field, SEED_TABLE, n_global, 1, _lib.p64(r)))
    if strong:
        first, stride, seed = rank, world, SEED_TABLE        # local j <- global j * world + rank of ONE table
    else:
        first, stride, seed = 0, 1, SEED_TABLE + rank        # an own table per rank
    t, table, out = fold_leg(env, zk, n, seed, first, stride, r, args.steps, args.warmup)
    dt, kern_ms = t["dt"], t["kernel_ms"]
    muls = half * args.steps * world
    algo_bytes = 96.0 * half                               # per launch: 2 x 32 B read + 32 B write per mul
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
    if world == 1:
        per = f"2^{args.log_n}-entry table on one GPU"
    elif strong:
        per = f"ONE 2^{args.log_n}-entry table low-bit-sharded {world}-way (config 5), 2^{args.log_n}/{world} entries per GPU"
    else:
        per = f"a 2^{args.log_n}-entry table per GPU"
    result = {
        "metric": f"field-mul/s (2^{args.log_n} MLE fold)", "value": muls / dt, "unit": "field-mul/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "prewarm_s": t["prewarm_s"], "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": f"{args.log_n}-variable MLE fold (partial_evaluate var 0), BLS12-381 Fr, {per}",
                   "log_n": args.log_n, "field": "bls12_381_fr", "entries_per_gpu": n,
                   "arithmetic": "255-bit Montgomery field, 8 x u32 limbs in HBM, products as 29-bit-limb v_mad_u64_u32 scans",
                   "sharding": "low-bit shard per rank, no data-path collective in the fold; the sharded proof and MSM of the same table are under config5_strong"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": None, "kernel": "fold0_kernel<Fr381>", "kernel_ms": kern_ms, "kernel_ms_median": t["kernel_ms_median"],
                     "achieved_median": algo_bytes / (t["kernel_ms_median"] * 1e-3) / 1e9, "timing_chunks": t["chunks"],
                     "algorithmic_bytes_per_launch": algo_bytes, "per": "GPU (rank 0's launches)"},
        "collectives": env.collectives,
    }
    failed = []
    for rnd in ("r4", "r3", "r2", "r1"):                     # PMC passes are separate rocprofv3 runs (committed summary)
        pmc = os.path.join(ROOT, "profiles", rnd, "fold_2p24_pmc.json")
        if n == 1 << 24 and os.path.exists(pmc):
            with open(pmc) as f:
                result["roofline"]["traffic"] = json.load(f)["hbm_bytes_per_launch"]
            result["roofline"]["traffic_source"] = (f"profiles/{rnd}/fold_2p24_pmc.json (committed summary of separate rocprofv3 --pmc FETCH_SIZE / "
                                                    "WRITE_SIZE passes, FETCH doubled per the gfx950 note; not re-measured in this run)")
            break
    if rank == 0 and not args.no_cpu_baseline:
        # what was timed is a correct fold: sampled outputs of the last launch against the oracle on the host mirror of the inputs
        result["post_check"] = post_check_fold(zk, field, out, r, seed, first, stride, half)
    del table, out
    watchdog = None
    if world > 1 and (env.tgroup is None or (rank == 0 and "ZK_BENCH_LEG_DEADLINE" in os.environ)):   # rehearsals: only when asked for (tests)
        watchdog = secondary_legs_watchdog(env, args, result)
    if world > 1 and strong and not args.no_weak:            # secondary: the weak-scaling form (a 2^log-n table per GPU)
        wt, wtab, wout = fold_leg(env, zk, n_global, SEED_TABLE + rank, 0, 1, r, min(args.steps, 300), min(args.warmup, 20), prewarm_s=0.0)
        wb = 96.0 * (n_global // 2) / (wt["kernel_ms"] * 1e-3) / 1e9
        result["weak"] = {"what": f"a 2^{args.log_n}-entry table per GPU, no data-path collective (linear by construction)",
                          "value": (n_global // 2) * min(args.steps, 300) * world / wt["dt"], "unit": "field-mul/s", "steps": min(args.steps, 300),
                          "kernel_ms": wt["kernel_ms"], "hbm_GBps_per_gpu": wb, "frac_per_gpu": wb / HBM_PEAK_GBPS}
        del wtab, wout
    if world > 1:
        try:
            result["sharded_sumcheck"] = sharded_gkr_sumcheck_leg(zk, env)
        except Exception as e:                             # noqa: BLE001
            result["sharded_sumcheck"] = {"error": repr(e)}
            failed.append("sharded_sumcheck")
    msm_shared = None
    if not args.no_config5:
        try:
            result["config5_strong"], msm_shared = config5_leg(zk, env, args)
        except Exception as e:                             # noqa: BLE001
            if world == 1:
                raise
            result["config5_strong"] = {"error": repr(e)}
            failed.append("config5_strong")
    if not args.no_msm:
        try:
            result["msm"] = msm_leg(zk, env, args, msm_shared)
            if rank == 0 and not args.no_cpu_baseline:
                result["msm"]["cpu_baseline"] = cpu_baseline_msm(zk)
        except Exception as e:                             # noqa: BLE001
            if world == 1:
                raise
            result["msm"] = {"error": repr(e)}
            failed.append("msm")
    if world == 1 and not args.no_configs:
        result["configs"] = baseline_configs(zk, args)
        result["paths"] = other_paths(zk, args)
    if rank == 0 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(zk, field)
    if failed:
        result["failed_legs"] = failed
    if watchdog:
        watchdog.cancel()
    return result


def secondary_legs_watchdog(env, args, result):
    """N > 1, one process per GPU: the headline (the sharded fold, no data-path collective) is measured; the legs that follow run the
    provers' collectives.  An exception in one of them is caught and reported (`failed_legs`); a HANG inside a collective would take the
    headline with it.  After ZK_BENCH_LEG_DEADLINE seconds (default 600) rank 0 prints the line with what has finished and every rank
    leaves (exit code 0, or 3 under --require-rccl)."""
    headline = json.dumps(dict(result, failed_legs=["watchdog: the legs after the headline did not finish in time (none reported)"]))
    deadline = float(os.environ.get("ZK_BENCH_LEG_DEADLINE", "600"))

    def fire():
        if env.rank == 0:
            try:
                done = dict(result)
                done["failed_legs"] = list(done.get("failed_legs", [])) + [f"watchdog: a leg after {sorted(k for k in done if k in ('weak', 'sharded_sumcheck', 'config5_strong', 'msm')) or 'the headline'} did not finish within {deadline:.0f} s"]
                line = json.dumps(done)
            except Exception:                               # noqa: BLE001  (the main thread was adding a key)
                line = headline
            print(line, flush=True)
        os._exit(3 if args.require_rccl else 0)

    t = threading.Timer(deadline, fire)
    t.daemon = True
    t.start()
    return t


# ---- checks of what was timed ---------------------------------------------------------------------------------------------
def verifier_equations_basic(zk, claimed, rp, ch):
    """verifier.rs:47-64 on a basic-sumcheck proof: every round's two half sums add up to the running claim; -> (ok, last claim)"""
    import numpy as np
    from zkmle_amd import _lib
    S = zk.sharded
    S._declare_host()
    Lb = zk.lib()
    cur, ok = claimed, True
    for k in range(rp.shape[0]):
        ok = ok and np.array_equal(S.fe_add(0, rp[k, 0], rp[k, 1]), cur)
        d = np.zeros(4, np.uint64)
        _lib.check(Lb.zk_fe_sub(0, _lib.p64(rp[k, 1]), _lib.p64(rp[k, 0]), _lib.p64(d)))
        m = np.zeros(4, np.uint64)
        _lib.check(Lb.zk_fe_mul(0, _lib.p64(np.ascontiguousarray(ch[k])), _lib.p64(d), _lib.p64(m)))
        cur = S.fe_add(0, rp[k, 0], m)
    return bool(ok), cur


def msm_identity_check(zk, scalars, a_int, d_int, got):
    """The bases are P_i = [a + i d] G, so MSM = [a sum s_i + d sum i s_i] G: two exact integer sums over the canonical scalars
    (16-bit pieces, vectorised) and ONE scalar multiplication by the checker (oracle) -- O(N), independent of the bucket method."""
    import numpy as np
    from oracle import oracle as O
    from zkmle_amd import _lib as L
    mont = scalars.evaluated_values
    n = mont.shape[0]
    canon = np.zeros_like(mont)
    L.check(L.lib().zk_vec_to_canonical(0, L.p64(mont), n, L.p64(canon)))
    idx = np.arange(n, dtype=np.uint64)
    s_sum, is_sum = 0, 0
    step = 1 << 20
    for k in range(4):
        for j in range(4):
            piece = (canon[:, k] >> np.uint64(16 * j)) & np.uint64(0xFFFF)
            shift = 64 * k + 16 * j
            s_sum += int(piece.sum(dtype=np.uint64)) << shift
            acc = 0
            for lo in range(0, n, step):                                        # 2^20 terms below 2^40 each: < 2^60
                acc += int((idx[lo:lo + step] * piece[lo:lo + step]).sum(dtype=np.uint64))
            is_sum += acc << shift
    R = O.modulus(O.FR381)
    k = (a_int * s_sum + d_int * is_sum) % R
    want = O.g1_mul_fr(O.g1_generator(), O.from_ints(O.FR381, [k])[0])
    return bool(np.array_equal(np.asarray(got, np.uint64).reshape(-1), np.asarray(want, np.uint64).reshape(-1)))


# ---- config 5 -----------------------------------------------------------------------------------------------------------------
def config5_leg(zk, env, args):
    """BASELINE config 5 at this rank count (strong scaling): the 2^log-n table low-bit-sharded over the ranks through the
    sharded sumcheck prover (include/zkmle.h zk_sharded_sumcheck_basic_prove, whole-table absorb off) and the 2^log-n-term MSM
    slice-sharded (zk_sharded_msm_g1).  Both are checked after the timing."""
    import numpy as np
    from zkmle_amd import _lib
    S = zk.sharded
    Lb = zk.lib()
    comm, rank, world = env.comm, env.rank, env.world
    n_global = 1 << args.log_n
    n = n_global // world
    out = {"what": f"one 2^{args.log_n}-entry Fr table and one 2^{args.log_n}-term MSM sharded {world}-way", "backend": env.collectives["backend"],
           "library_comm": env.collectives["library"]}
    # -- sumcheck prover rounds on the sharded table
    table = zk.MultilinearPolynomial.alloc(0, n)
    _lib.check(Lb.zk_table_fill_random_strided(table._h, SEED_TABLE, rank, world))
    shard = S.GpuShard(table)
    for _ in range(200):                                                      # untimed (the same count on every rank: the proofs are collective):
        S.sumcheck_basic_prove_device(comm, shard, absorb_table=False)        # a cold GPU under-clocks its first tens of milliseconds
    rx0, nc0 = comm.native_stats()
    nproofs = 20
    dt, (claimed, rp, ch) = env.timed(lambda: S.sumcheck_basic_prove_device(comm, shard, absorb_table=False), nproofs)
    rx1, nc1 = comm.native_stats()
    # verifier equations (verifier.rs:47-70) on the proof just timed: claim chain + the table evaluated at the challenges
    ok, cur = verifier_equations_basic(zk, claimed, rp, ch)
    ok = ok and np.array_equal(S.mle_evaluate(comm, table, ch), cur)
    # unavoidable HBM bytes of the round phase: the segment sums read the table once (32 B / entry); every pass of m rounds reads what is
    # left again and writes 2^-m of it, down to the 2^11 entries of the one-workgroup tail (passes as csrc/zkmle_sumcheck.hip spreads them:
    # at most 7 rounds each, evenly)
    left, cur, traffic = args.log_n - 11, float(n_global), 32.0 * n_global
    while left > 0:
        passes = (left + 6) // 7
        m = (left + passes - 1) // passes
        traffic += 32.0 * cur * (1 + 2.0 ** -m)
        cur /= 2 ** m
        left -= m
    out["sumcheck"] = {"what": f"Prover::prove rounds of the 2^{args.log_n} table ({rp.shape[0]} rounds), 2^{args.log_n}/{world} entries per rank, "
                               "up to 7 rounds per pass over the shard (13 = 7 + 6): one all-reduce(SUM) of 2^m x 9 int64 words (the segment sums that carry the pass's m rounds) on the prover's stream per pass -- at one rank the pass's last workgroup runs the exchange itself --, replicated one-launch tail, transcript steps on each rank's host through the mailbox",
                       "ms_per_proof": dt * 1e3, "field_mul_per_s": (n_global - 1) / dt, "rounds": int(rp.shape[0]),
                       "proofs_timed": nproofs, "collectives_per_proof": (nc1 - nc0) // nproofs, "bytes_received_per_proof": (rx1 - rx0) // nproofs,
                       "hbm_GBps_whole_proof": traffic / dt / 1e9, "frac_of_hbm_peak_whole_proof": traffic / dt / 1e9 / (HBM_PEAK_GBPS * world),
                       "verifier_equations_hold": bool(ok)}
    if not ok:
        raise RuntimeError("config 5: the timed sharded proof fails the verifier's equations")
    del shard, table
    # -- MSM, terms sliced over the ranks: rank g owns the terms [g n / G, (g + 1) n / G)
    lo = rank * n
    a = zk.from_ints(0, [SEED_MSM])[0]
    d = zk.from_ints(0, [MSM_D])[0]
    a_lo = zk.from_ints(0, [SEED_MSM + lo * MSM_D])[0]                        # P_i = [a + i d] G for the global index i
    bases = zk.G1Bases.synthetic(n, a_lo, d)
    scalars = zk.MultilinearPolynomial.alloc(0, n)
    _lib.check(Lb.zk_table_fill_random_strided(scalars._h, SEED_MSM, lo, 1))
    S.msm_device(comm, scalars, bases, 0, True)                               # warm-up
    dt, (pt, st) = env.timed(lambda: S.msm_device(comm, scalars, bases, 0, True), args.msm_reps)
    same = np.array_equal(comm.all_gather(pt), np.broadcast_to(pt, (world, 12)))
    W = st["windows"]
    out["msm"] = {"what": f"2^{args.log_n}-term MSM, 2^{args.log_n}/{world} terms per rank, one all-gather of {world} affine points + {world - 1} additions",
                  "ms_per_msm": dt * 1e3, "g1_add_per_s": W * n_global / dt, "terms_per_s": n_global / dt, "window_bits": st["window_bits"],
                  "windows": W, "local_device_ms": st["ms_total"], "same_point_on_every_rank": bool(same),
                  "point_x_limb0": int(pt[0])}
    if rank == 0:                                                             # the O(N) identity on ALL the scalars of the global MSM
        if world > 1:
            full = zk.MultilinearPolynomial.alloc(0, n_global)
            _lib.check(Lb.zk_table_fill_random_strided(full._h, SEED_MSM, 0, 1))
        else:
            full = scalars
        t0 = time.perf_counter()
        good = msm_identity_check(zk, full, SEED_MSM, MSM_D, pt)
        out["msm"]["post_check"] = {"identity": "MSM == [a sum s_i + d sum i s_i] G for bases [a + i d] G", "holds": good,
                                    "check_s": time.perf_counter() - t0}
        if not good:
            raise RuntimeError("config 5: the timed MSM's point fails the linear identity of its bases")
    # the same MSM on precomputed window-shifted bases (zk_g1_bases_precompute, once per setup): ONE bucket set, 22-bit windows
    t0 = time.perf_counter()
    pre_c = bases.precompute(0)
    build_s = time.perf_counter() - t0
    S.msm_device(comm, scalars, bases, 0, True)                               # warm-up
    dtp, (ptp, stp) = env.timed(lambda: S.msm_device(comm, scalars, bases, 0, True), args.msm_reps)
    Wp = stp["windows"]
    out["msm_precomputed"] = {"what": "the same MSM with one pre-converted copy of every base per window, 2^(c w) B_i: all windows feed one bucket set",
                              "ms_per_msm": dtp * 1e3, "g1_add_per_s": Wp * n_global / dtp, "terms_per_s": n_global / dtp, "window_bits": stp["window_bits"],
                              "windows": Wp, "local_device_ms": stp["ms_total"], "table_bytes_per_gpu": 128 * n * Wp, "precompute_s": build_s,
                              "same_point_as_plain": bool(np.array_equal(ptp, pt))}
    if pre_c != stp["window_bits"] or not np.array_equal(ptp, pt):
        raise RuntimeError("config 5: the MSM on precomputed bases disagrees with the plain MSM")
    shared = {"dt": dt, "pt": pt, "st": st, "check": out["msm"].get("post_check"), "pre": out["msm_precomputed"], "pre_st": stp} if world == 1 else None
    return out, shared


def sharded_gkr_sumcheck_leg(zk, env, log_local=20):
    """A whole GKR sumcheck (4 tables, degree 2; sumcheck_gkr_protocol.rs:24-67) sharded over the ranks, weak: 2^20 entries per
    table per rank.  One C-ABI call per proof (zk_sharded_sumcheck_gkr_prove)."""
    import numpy as np
    S = zk.sharded
    rank, world, comm = env.rank, env.world, env.comm
    n = 1 << log_local
    MP = zk.MultilinearPolynomial
    tabs = [[MP.random(0, n, 0x5EED0400 + 16 * rank + 2 * p + f) for f in range(2)] for p in range(2)]
    shard = S.GpuSumShard(0, tabs)
    claimed = np.zeros(4, np.uint64)
    S.sumcheck_gkr_prove_device(comm, shard, claimed, zk.Transcript())      # warm-up
    dt, (co, ch, fin) = env.timed(lambda: S.sumcheck_gkr_prove_device(comm, shard, claimed, zk.Transcript()), 3)
    rounds = int(co.shape[0])
    return {"what": f"GKR sumcheck on 4 tables of 2^{log_local} entries per rank ({rounds} rounds over {world} ranks), "
                    "one all-reduce(SUM) of 27 int64 words per large round", "ms_per_proof": dt * 1e3, "rounds": rounds,
            "ms_per_round": dt * 1e3 / rounds, "field_mul_per_s": 5.0 * 2 * n * world / dt, "backend": env.collectives["backend"]}


# ---- MSM ------------------------------------------------------------------------------------------------------------------------
def msm_leg(zk, env, args, shared=None):
    """G1-add/s on the 2^log-n Pippenger MSM (BASELINE.json's second metric): one 2^log-n-term MSM per rank, then one
    all-gather of `world` affine points + world-1 additions (no bandwidth-sized collective).  At N = 1 this IS config 5's MSM
    (same scalars, same bases): it is run and checked once and reported under both keys."""
    rank, world, comm = env.rank, env.world, env.comm
    n = 1 << args.log_n
    S = zk.sharded
    if shared is not None:
        dt, pt, st, check = shared["dt"], shared["pt"], shared["st"], shared["check"]
    else:
        a = zk.from_ints(0, [SEED_MSM + rank])[0]
        d = zk.from_ints(0, [MSM_D])[0]
        bases = zk.G1Bases.synthetic(n, a, d)                   # P_i = [a + i d] G, generated on the device
        scalars = zk.MultilinearPolynomial.random(0, n, SEED_MSM + 97 * rank)
        S.msm_device(comm, scalars, bases, 0, True)             # warm-up
        dt, (pt, st) = env.timed(lambda: S.msm_device(comm, scalars, bases, 0, True), args.msm_reps)
        check = None
        if world == 1:
            t0 = time.perf_counter()
            good = msm_identity_check(zk, scalars, SEED_MSM, MSM_D, pt)
            check = {"identity": "MSM == [a sum s_i + d sum i s_i] G for bases [a + i d] G", "holds": good, "check_s": time.perf_counter() - t0}
            if not good:
                raise RuntimeError("the timed MSM's point fails the linear identity of its bases")
    W = st["windows"]
    res = {"metric": f"G1-add/s (2^{args.log_n} MSM)", "value": W * n * world / dt, "unit": "G1-add/s", "g1_add_per_s": W * n * world / dt,
           "terms_per_s": n * world / dt, "ms_per_msm": dt * 1e3, "window_bits": st["window_bits"], "windows": W,
           "adds_per_term": W, "device_ms": {k: st[k] for k in ("ms_digits", "ms_sort", "ms_buckets", "ms_reduce", "ms_total")},
           "unique_bytes_per_term": 128, "hbm_GBps_unique": 128.0 * n / (st["ms_total"] * 1e-3) / 1e9,
           "workload": f"2^{args.log_n} random Fr scalars x synthetic affine bases [a + i d]G per GPU; one Pippenger per rank, all-gather of {world} points"}
    if check is not None:
        res["post_check"] = check
    if rank == 0:                                           # (one micro-benchmark child process per job, not per rank)
        try:
            res["roofline"] = msm_roofline(st)
        except Exception as e:                              # noqa: BLE001
            res["roofline"] = {"error": repr(e)}
    if shared is not None and shared.get("pre"):
        res["precomputed_bases"] = dict(shared["pre"], device_ms={k: shared["pre_st"][k] for k in ("ms_digits", "ms_sort", "ms_buckets", "ms_reduce", "ms_total")})
        try:
            res["precomputed_bases"]["roofline"] = msm_roofline(shared["pre_st"])
        except Exception as e:                              # noqa: BLE001
            res["precomputed_bases"]["roofline"] = {"error": repr(e)}
    return res


# v_mad_u64_u32 per lane of one mixed addition of msm_bucket_sum_kernel (csrc/g1u.cuh g1u_madd on 14 x 29-bit limbs of Fq381):
# 6 products (392) + 2 squarings (301) + 1 dual product with one reduction (588).  Source-level count = the ISA's common path
# (tools/count_mads.py, profiles/r2/msm_mad_count.txt).
MADS_PER_MIXED_ADD = 6 * 392 + 2 * 301 + 588
# v_mad_u64_u32 per pair index of the GKR round kernels on 2 products x 2 factors (ISA counts of the loop bodies, tools/kernel_isa_stats.py;
# source level: a fold by the uniform multiplier = 81 + 2 x 9, a raw product = 81; Fr381's p has a limb equal to 1, which saves a few):
# fused round = 8 folds + 4 raw products, first round = 6 raw products
MADS_FOLD_ROUND = 1100
MADS_ROUND = 486


def msm_roofline(st):
    """integer-VALU roofline of the dominant MSM kernel (msm_bucket_sum_kernel): achieved v_mad_u64_u32 lane-ops/s from the
    bucket phase's HIP-event time and the counted multiply-adds per mixed addition, against the rate the chip sustains on a
    register-resident v_mad_u64_u32 chain, re-measured in this run (tools/microbench, `mad_u64_u32`)."""
    adds = float(st["entries"])                             # one mixed addition per (term, window) entry with a non-zero digit
    achieved = adds * MADS_PER_MIXED_ADD / (st["ms_buckets"] * 1e-3)
    peak, src = measured_mad_peak()
    out = {"bound": "valu v_mad_u64_u32", "kernel": "msm_bucket_sum_kernel", "achieved": achieved / 1e12, "peak": peak / 1e12, "unit": "Tmad/s",
           "frac": achieved / peak, "mixed_adds": adds, "mads_per_mixed_add": MADS_PER_MIXED_ADD, "kernel_ms": st["ms_buckets"],
           "peak_source": src, "traffic": None}
    # the kernel's memory side, for completeness: HBM bytes per launch from the committed PMC passes of the same 2^24 workload (16-bit windows,
    # or one bucket set of 22-bit windows on precomputed bases), scaled by the additions of this launch
    name = "msm_2p24_c22pre_bucket_pmc.json" if st.get("window_bits", 16) > 16 else "msm_2p24_c16_bucket_pmc.json"
    pmc = os.path.join(ROOT, "profiles", "r4", name)
    if os.path.exists(pmc):
        with open(pmc) as f:
            d = json.load(f)
        per_add = d["hbm_bytes_per_launch"] / (d["algorithmic_bytes_per_launch"] / 128.0)
        out["traffic"] = per_add * adds
        out["traffic_GBps"] = per_add * adds / (st["ms_buckets"] * 1e-3) / 1e9
        out["traffic_over_128B_per_add"] = d["traffic_over_algorithmic"]
        out["traffic_source"] = f"profiles/r4/{name} (committed summary of separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over a 2^24-term MSM; bytes per mixed addition x this launch's additions)"
    return out


_MAD_PEAK = None
_MAD_LOCK = threading.Lock()


def measured_mad_peak():
    global _MAD_PEAK
    with _MAD_LOCK:
        if _MAD_PEAK is None:
            _MAD_PEAK = _measure_mad_peak()
    return _MAD_PEAK


def _measure_mad_peak():
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


# ---- BASELINE configs 2, 3, 4 (one GPU) -------------------------------------------------------------------------------------------
def event_time_ms(fn, reps, warm=3, blocker=None):
    """mean HIP-event time of `fn` (enqueue-only work on the current stream) over `reps` back-to-back calls.  `blocker` (enqueue-only
    too) is queued first and keeps the GPU busy while the host queues the timed launches behind it: a launch shorter than the host's
    enqueue time (~10 us through ctypes, 40 us on a loaded host) is then timed at the GPU's rate, not the host's."""
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if blocker:
        blocker()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def baseline_configs(zk, args):
    """BASELINE.json configs 2, 3 and 4 on this GPU, each timed and then checked (proofs through the verifier, the MSM through the
    identity of its bases).  A failed check aborts the run: no line."""
    import ctypes as C
    import numpy as np
    import torch
    from zkmle_amd import _lib
    L = zk.lib()
    MP = zk.MultilinearPolynomial
    stream = torch.cuda.current_stream().cuda_stream
    out = {}
    if not args.no_cpu_baseline:
        out["cfg1"] = config1(zk)
    # -- config 2: 20-variable sumcheck fold + the whole Prover::prove (prover.rs:35-71)
    n = 1 << 20
    poly = MP.random(0, n, 0x5EED0002)
    dst = MP.alloc(0, n // 2)
    r = np.zeros(4, np.uint64)
    _lib.check(L.zk_host_fill_random(0, SEED_TABLE, 77, 1, _lib.p64(r)))
    big, big_dst = MP.random(0, 1 << 24, 0x5EED0005), MP.alloc(0, 1 << 23)

    def busy():                                             # ~8 ms of GPU work in front of the 400 timed 10-us launches
        for _ in range(60):
            _lib.check(L.zk_mle_fold(big._h, 0, _lib.p64(r), big_dst._h, stream))
    fold_ms = min(event_time_ms(lambda: _lib.check(L.zk_mle_fold(poly._h, 0, _lib.p64(r), dst._h, stream)), 400, warm=200, blocker=busy)
                  for _ in range(3))                    # the smallest of three: one run on a loaded host read 156 us for this 11-us launch
    del big, big_dst
    zk.Prover.init(0, poly).prove()                          # (a Prover proves once: prove() appends to its own transcript, prover.rs:10,38-58)
    prover = zk.Prover.init(0, poly)
    t0 = time.perf_counter()
    proof = prover.prove()
    prove_s = time.perf_counter() - t0
    st = zk.sumcheck.last_stats()
    verified = zk.Verifier.init().verify(proof)
    if not verified:
        raise SystemExit("bench.py: config 2's proof is rejected by the verifier")
    gb = 96.0 * (n // 2) / (fold_ms * 1e-3) / 1e9
    out["cfg2"] = {"what": "20-variable MLE sumcheck fold, BLS12-381 Fr, 1xMI355X", "fold_us": fold_ms * 1e3, "fold_field_mul_per_s": (n // 2) / (fold_ms * 1e-3),
                   "fold_GBps": gb, "fold_frac": gb / HBM_PEAK_GBPS, "fold_note": "the 2^20 table (32 MiB in + 16 MiB out) fits the 256 MiB MALL: above-HBM rates are cache hits",
                   "prove_ms": prove_s * 1e3, "rounds_ms": st["ms_rounds"], "absorb_ms": st["ms_absorb"],
                   "absorb_GBps": 32.0 * n / (st["ms_absorb"] * 1e-3) / 1e9,
                   "absorb_note": "prove_ms is the reference's protocol, not a GPU number: prover.rs:38-39 absorbs the whole table into ONE sequential Keccak sponge (one host core; the GPU converts to canonical big-endian bytes beside it)",
                   "rounds_field_mul_per_s": (n - 1) / (st["ms_rounds"] * 1e-3), "post_check": {"verifier_accepts": bool(verified)}}
    del poly, dst, prover, proof
    # -- config 3: multilinear_kzg::commit, 2^20-scalar Pippenger MSM (multilinear_kzg.rs:25-45)
    bases = zk.G1Bases.synthetic(n, zk.from_ints(0, [SEED_MSM])[0], zk.from_ints(0, [MSM_D])[0])
    scalars = MP.random(0, n, SEED_MSM + 3)
    zk.kzg.msm(scalars, bases, 0, True)
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        pt, ms = zk.kzg.msm(scalars, bases, 0, True)
        wall = time.perf_counter() - t0
        if best is None or wall < best[0]:
            best = (wall, ms)
    wall, ms = best
    good = msm_identity_check(zk, scalars, SEED_MSM, MSM_D, pt)
    if not good:
        raise SystemExit("bench.py: config 3's MSM fails the linear identity of its bases")
    out["cfg3"] = {"what": "multilinear_kzg::commit 2^20-scalar Pippenger MSM, BLS12-381 G1, 1xMI355X", "ms_per_msm": wall * 1e3,
                   "device_ms": {k: ms[k] for k in ("ms_digits", "ms_sort", "ms_buckets", "ms_reduce", "ms_total")},
                   "terms_per_s": n / wall, "g1_add_per_s": ms["windows"] * n / wall, "window_bits": ms["window_bits"], "windows": ms["windows"],
                   "post_check": {"identity_holds": bool(good)}}
    t0 = time.perf_counter()
    pre_c = bases.precompute(0)                              # once per setup: window-shifted copies, one bucket set
    build_s = time.perf_counter() - t0
    zk.kzg.msm(scalars, bases, 0, True)
    bestp = None
    for _ in range(5):
        t0 = time.perf_counter()
        ptp, msp = zk.kzg.msm(scalars, bases, 0, True)
        wallp = time.perf_counter() - t0
        if bestp is None or wallp < bestp[0]:
            bestp = (wallp, msp)
    if not np.array_equal(ptp, pt) or bestp[1]["window_bits"] != pre_c:
        raise SystemExit("bench.py: config 3's MSM on precomputed bases disagrees with the plain one")
    out["cfg3"]["precomputed_bases"] = {"ms_per_msm": bestp[0] * 1e3, "window_bits": pre_c, "windows": bestp[1]["windows"], "terms_per_s": n / bestp[0],
                                        "device_ms": {k: bestp[1][k] for k in ("ms_digits", "ms_sort", "ms_buckets", "ms_reduce", "ms_total")},
                                        "precompute_s": build_s, "table_bytes": 128 * n * bestp[1]["windows"], "same_point_as_plain": True}
    del bases, scalars
    # -- config 4: GKR prover, depth 3, 2^22 gates per layer (gkr_protocol.rs:57-143 on gate lists)
    lg, depth = 22, 3
    ng = 1 << lg
    rng = np.random.default_rng(0x5EED0004)
    rows = []
    for _ in range(depth):
        g = np.zeros((ng, 4), np.uint64)
        g[:, 0] = rng.integers(0, ng, ng)
        g[:, 1] = rng.integers(0, ng, ng)
        g[:, 2] = np.arange(ng)
        g[:, 3] = rng.integers(0, 2, ng)
        rows.append(g)
    x = MP.random(0, ng, 0x5EED0004).evaluated_values
    t0 = time.perf_counter()
    circuit = zk.gkr.SparseCircuit(rows, [lg] * depth, ng)
    compile_s = time.perf_counter() - t0
    zk.gkr.sparse_prove(0, None, None, x, circuit=circuit)
    t0 = time.perf_counter()
    proof = zk.gkr.sparse_prove(0, None, None, x, circuit=circuit)
    prove_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    ok = zk.gkr.sparse_verify(0, rows, [lg] * depth, proof, x)
    verify_s = time.perf_counter() - t0
    if not ok:
        raise SystemExit("bench.py: config 4's proof is rejected by the sparse verifier")
    del circuit, rows
    # the two round kernels of sumcheck_gkr_protocol.rs:113-143 alone, on 4 tables of 2^22 (2 products x 2 factors): enqueue-only calls
    S = zk.sharded
    lib = S._declare_host()
    sc = zk.sumcheck._decl()
    tabs = [MP.random(0, ng, 0x5EED0440 + k) for k in range(4)]
    outs = [MP.alloc(0, ng // 2) for _ in range(4)]
    ta = (C.c_void_p * 4)(*[t._h for t in tabs])
    oa = (C.c_void_p * 4)(*[t._h for t in outs])

    def fold_round():
        _lib.check(lib.zk_sumpoly_fold_round_evals(ta, oa, 2, 2, _lib.p64(r), None))

    fre_ms = event_time_ms(fold_round, 200, warm=20)
    re_ms = event_time_ms(lambda: _lib.check(sc.zk_sumpoly_round_evals(ta, 2, 2, None)), 200, warm=20)
    q = ng // 4
    fre_bytes = 4 * (ng + ng // 2) * 32.0
    re_bytes = 4 * ng * 32.0
    mad_peak, mad_src = measured_mad_peak()
    absorb_s = prove_s - sum(float(v) for v in proof.ms_layers) * 1e-3
    out["cfg4"] = {"what": "GKR prover, depth-3 layered circuit, 2^22 gates/layer (random wiring), BLS12-381 Fr, 1xMI355X; sparse (linear-time) prover",
                   "circuit_compile_s": compile_s, "prove_s": prove_s, "device_ms_per_layer": [float(v) for v in proof.ms_layers], "verify_s": verify_s,
                   "gates_per_s": depth * ng / prove_s,
                   "absorb_s": absorb_s, "absorb_GBps": 32.0 * ng / absorb_s / 1e9,
                   "absorb_note": "prove_s minus the layers' device time: the sequential host Keccak absorb of the 2^22-entry output layer (gkr_protocol.rs:49) -- the reference's protocol, one host core, not a GPU number",
                   "round_kernels": {
                       "what": "4 tables of 2^22 entries (2 products x 2 factors), HIP events over 200 back-to-back enqueue-only launches",
                       "fold_round_evals_kernel": {"us": fre_ms * 1e3, "GBps": fre_bytes / (fre_ms * 1e-3) / 1e9, "frac": fre_bytes / (fre_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                                   "algorithmic_bytes": fre_bytes, "field_mul_per_s": 12.0 * q / (fre_ms * 1e-3),
                                                   "field_mul_per_pair_index": "8 fold products + 4 evaluation products (nodes 0 and infinity; the node 1 is derived)",
                                                   "mads_per_pair_index": MADS_FOLD_ROUND, "Tmad_per_s": MADS_FOLD_ROUND * q / (fre_ms * 1e-3) / 1e12,
                                                   "frac_valu": MADS_FOLD_ROUND * q / (fre_ms * 1e-3) / mad_peak},
                       "round_evals_kernel": {"us": re_ms * 1e3, "GBps": re_bytes / (re_ms * 1e-3) / 1e9, "frac": re_bytes / (re_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                              "algorithmic_bytes": re_bytes, "field_mul_per_s": 6.0 * (ng // 2) / (re_ms * 1e-3),
                                              "field_mul_per_pair_index": "6 evaluation products (nodes 0, 1, infinity of 2 products)",
                                              "mads_per_pair_index": MADS_ROUND, "Tmad_per_s": MADS_ROUND * (ng // 2) / (re_ms * 1e-3) / 1e12,
                                              "frac_valu": MADS_ROUND * (ng // 2) / (re_ms * 1e-3) / mad_peak},
                       "frac_valu_note": "v_mad_u64_u32 per pair index (ISA count, tools/kernel_isa_stats.py: profiles/r4/gkr_round_kernels_isa.md) x pair indices / time, against the "
                                         "in-run multiply-add peak (" + mad_src + "); multiply-adds are ~40 % of these kernels' VALU instructions, so frac_valu understates how busy the VALU pipe is"},
                   "post_check": {"sparse_verifier_accepts": bool(ok)},
                   "note": "prove_s includes the sequential host Keccak absorb of the 2^22-entry output layer; circuit_compile_s is paid once per circuit"}
    return out


def config1(zk):
    """BASELINE config 1: the reference's own bench shape (sumcheck_protocol/benches/basic_sumcheck_benchmark.rs: one iteration = Prover::init +
    prove AND Verifier::verify) on a 12-variable BLS12-381 Fr table, the oracle on ONE host thread; the product's proof of the same table
    beside it (it must be the same bytes)."""
    import numpy as np
    from oracle import oracle as O
    from zkmle_amd import _lib
    n = 1 << 12
    tab = np.zeros((n, 4), np.uint64)
    _lib.check(zk.lib().zk_host_fill_random(0, SEED_TABLE, 0, n, _lib.p64(tab)))
    cs, rp, _ = O.sumcheck_basic_prove(0, tab)
    O.sumcheck_basic_verify(0, tab, cs, rp)
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 2.0:                   # ~2 s of single-core work
        cs, rp, _ = O.sumcheck_basic_prove(0, tab)
        okr = O.sumcheck_basic_verify(0, tab, cs, rp)
        reps += 1
    cpu_ms = (time.perf_counter() - t0) / reps * 1e3
    zk.Verifier.init().verify(zk.Prover.init(0, tab).prove())
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        proof = zk.Prover.init(0, tab).prove()
        ok = zk.Verifier.init().verify(proof)
        ts.append((time.perf_counter() - t0) * 1e3)
    same = bool(np.array_equal(proof.initial_claimed_sum, cs) and np.array_equal(proof.round_univariate_polynomials, rp))
    if not (ok and okr and same):
        raise SystemExit("bench.py: config 1: the GPU proof of the 2^12 table differs from the oracle's or is rejected")
    return {"what": "sumcheck_protocol bench shape (init + prove + verify) on a 12-variable BLS12-381 Fr table", "cpu_reference_ms": cpu_ms,
            "cpu": {"kind": "port", "cores": 1, "iterations": reps, "sample": "the oracle's restatement of prover.rs / verifier.rs, one host thread"},
            "gpu_ms": sorted(ts)[len(ts) // 2], "gpu_note": "host upload of the 128 KiB table + proof + verifier's evaluate per iteration: latency-bound (12 rounds), the table is too small to occupy the chip",
            "post_check": {"gpu_proof_equals_oracle_proof": same, "verifier_accepts": bool(ok)}}


def other_paths(zk, args):
    """The other functions `north_star` names, at BASELINE's largest size on this GPU, each timed and then checked: evaluate (evaluation_form.rs:21-33),
    trusted setup (trusted_setup.rs:24-60), commit + open_and_prove (multilinear_kzg.rs:25-126; the opening that was timed goes through the pairing
    check :131-158), dense gkr::prove on the largest reference-shaped circuit whose dense wiring tables are kept (gkr_protocol.rs:26-143)."""
    import random
    import numpy as np
    import torch
    from zkmle_amd import _lib
    L = zk.lib()
    MP = zk.MultilinearPolynomial
    out = {}

    def sync():
        torch.cuda.synchronize()

    lg = args.log_n
    n = 1 << lg
    # -- evaluate: n - 1 field multiplications, 96 (n - 1) algorithmic bytes
    poly = MP.random(0, n, 0x5EED0002)
    point = MP.random(0, lg, 77).evaluated_values
    for _ in range(3):
        poly.evaluate(point)
    ts = []
    for _ in range(20):
        sync(); t0 = time.perf_counter()
        v = poly.evaluate(point)
        ts.append(time.perf_counter() - t0)
    ev_s = sorted(ts)[len(ts) // 2]
    cur = poly                                              # the check: the same value as `lg` successive partial_evaluate launches (the fold kernel)
    for i in range(lg):
        cur = MP.partial_evaluate(cur, 0, point[i])
    chain_ok = bool(np.array_equal(np.asarray(v).reshape(-1), cur.evaluated_values.reshape(-1)))
    if not chain_ok:
        raise SystemExit("bench.py: evaluate() differs from the chain of partial_evaluate launches")
    floor = 96.0 * (n - 1)
    out[f"evaluate_2p{lg}"] = {"what": f"MultilinearPolynomial::evaluate on a 2^{lg}-entry Fr table, one call (host wall time, value downloaded)", "ms": ev_s * 1e3,
                               "field_mul_per_s": (n - 1) / ev_s, "floor_bytes": floor, "GBps_vs_floor": floor / ev_s / 1e9, "frac_vs_floor": floor / ev_s / 1e9 / HBM_PEAK_GBPS,
                               "note": "the floor is the reference's schedule (one pass per variable: 96 B per multiplication); the product folds up to 4 variables per pass, "
                                       "so it moves fewer bytes than the floor counts and can exceed 1.0 of it",
                               "post_check": {"equals_chain_of_partial_evaluate": chain_ok}}
    del cur
    # -- trusted setup, commit, open_and_prove, verify
    taus = zk.from_ints(0, [0x1000003 * (i + 1) + 12345 for i in range(lg)])
    opening = zk.from_ints(0, [0x2000003 * (i + 7) + 999 for i in range(lg)])
    sync(); t0 = time.perf_counter()
    setup = zk.TrustedSetup.initialize_setup(taus)
    sync(); setup_s = time.perf_counter() - t0
    out[f"setup_2p{lg}"] = {"what": f"TrustedSetup::initialize_setup for {lg} variables: compute_lagrange_basis + 2^{lg} fixed-base [L_i(tau)]G + batch to affine + {lg} G2 powers (host)",
                            "s": setup_s, "points_per_s": n / setup_s}
    zk.MultilinearKZG.commit_to_polynomial(poly, setup)
    sync(); t0 = time.perf_counter()
    com = zk.MultilinearKZG.commit_to_polynomial(poly, setup)
    sync(); commit_s = time.perf_counter() - t0
    sync(); t0 = time.perf_counter()
    setup.opening_key()
    sync(); key_s = time.perf_counter() - t0
    zk.MultilinearKZG.open_and_prove(poly, setup, opening)
    opens = []
    for _ in range(3):
        sync(); t0 = time.perf_counter()
        prf = zk.MultilinearKZG.open_and_prove(poly, setup, opening)
        sync(); opens.append(time.perf_counter() - t0)
    open_s = min(opens)
    t0 = time.perf_counter()
    verified = bool(zk.MultilinearKZG.verify(setup, com, opening, prf))
    verify_s = time.perf_counter() - t0
    same_eval = bool(np.array_equal(np.asarray(prf.evaluation).reshape(-1), np.asarray(poly.evaluate(opening)).reshape(-1)))
    if not (verified and same_eval):
        raise SystemExit("bench.py: the KZG opening that was timed fails the pairing check")
    out[f"kzg_commit_2p{lg}"] = {"what": "commit_to_polynomial on a real setup ([L_i(tau)]G bases)", "ms": commit_s * 1e3, "terms_per_s": n / commit_s}
    out[f"kzg_open_2p{lg}"] = {"what": f"open_and_prove: evaluate + {lg} quotient MSMs of 2^{lg - 1} .. 1 terms on pre-summed bases (opening key built once per setup: {key_s * 1e3:.1f} ms)",
                               "ms": open_s * 1e3, "terms_per_s": (n - 1) / open_s, "opening_key_ms": key_s * 1e3,
                               "post_check": {"pairing_check_of_the_timed_opening": verified, "verify_s": verify_s, "pairings": lg + 1, "evaluation_equals_evaluate": same_eval}}
    del setup, prf, poly
    # -- dense gkr::prove, the reference's own shape (layer i: 2^i gates reading 2^(i+1) wires; dense add_i / mul_i of 2^(3i+2) entries)
    depth = 8
    rng = random.Random(8)
    layers = []
    for i in range(depth):
        n_in = 1 << (i + 1)
        layers.append(zk.gkr.Layer([zk.gkr.Gate(rng.randrange(n_in), rng.randrange(n_in), o, rng.choice([0, 1])) for o in range(1 << i)]))
    circuit = zk.gkr.Circuit(0, layers)
    x = MP.random(0, 1 << depth, 0x5EED0008).evaluated_values
    zk.gkr.prove(circuit, x)
    ts = []
    for _ in range(3):
        sync(); t0 = time.perf_counter()
        gp = zk.gkr.prove(circuit, x)
        sync(); ts.append(time.perf_counter() - t0)
    ok = zk.gkr.verify(circuit, gp, x)
    if not ok:
        raise SystemExit("bench.py: the dense GKR proof that was timed is rejected by gkr_protocol::verify")
    # the same proof on the reference's dense add_i / mul_i tables (ZK_GKR_DENSE_TABLES=1, read per call): must be the same bytes
    os.environ["ZK_GKR_DENSE_TABLES"] = "1"
    try:
        zk.gkr.prove(circuit, x)
        td = []
        for _ in range(3):
            sync(); t0 = time.perf_counter()
            gd = zk.gkr.prove(circuit, x)
            sync(); td.append(time.perf_counter() - t0)
    finally:
        del os.environ["ZK_GKR_DENSE_TABLES"]
    same = bool(all(np.array_equal(u, v) for u, v in zip(gp._flat, gd._flat)) and np.array_equal(gp.claimed_sum, gd.claimed_sum)
                and np.array_equal(gp.wb_evaluations, gd.wb_evaluations) and np.array_equal(gp.wc_evaluations, gd.wc_evaluations))
    if not same:
        raise SystemExit("bench.py: gkr_protocol::prove from the gate lists and on the dense tables differ")
    out["gkr_dense"] = {"what": f"gkr_protocol::prove, reference-shaped circuit of depth {depth} (2^{depth} inputs, layer i: 2^i gates), BLS12-381 Fr; proved from the gate lists "
                                f"(one stream of kernels, lists compiled once per circuit), the reference's transcript and bytes", "ms": min(ts) * 1e3, "rounds": sum(2 * (i + 1) for i in range(depth)),
                        "dense_tables_ms": min(td) * 1e3,
                        "dense_tables_note": f"the same call with ZK_GKR_DENSE_TABLES=1: the reference's representation (the last layer's add_i / mul_i have 2^{3 * (depth - 1) + 2} entries each, "
                                             f"its f(b,c) tables 2^{2 * (depth - 1) + 2})",
                        "post_check": {"verifier_accepts": bool(ok), "both_representations_same_bytes": same},
                        "note": "circuits of BASELINE config 4's size go through the same gate-list prover (configs.cfg4)"}
    # -- prove_succinct (succinct_gkr_protocol.rs:35-169): the same proof + commitment to the input layer + two KZG openings at the last layer's challenges
    taus8 = MP.random(0, 1 << 4, 0x5EED0009).evaluated_values[:depth]
    setup8 = zk.TrustedSetup.initialize_setup(taus8)
    zk.gkr.prove_succinct(circuit, x, setup8)
    tsu = []
    for _ in range(3):
        sync(); t0 = time.perf_counter()
        sp = zk.gkr.prove_succinct(circuit, x, setup8)
        sync(); tsu.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    oks = bool(zk.gkr.verify_succinct(circuit, sp, setup8))
    vs = time.perf_counter() - t0
    if not oks:
        raise SystemExit("bench.py: the succinct GKR proof that was timed is rejected by verify_succinct")
    out["gkr_succinct"] = {"what": f"succinct_gkr_protocol::prove_succinct on the same circuit: gkr::prove + commit_to_polynomial of the 2^{depth} inputs + two open_and_prove",
                           "ms": min(tsu) * 1e3, "post_check": {"verify_succinct_accepts": oks, "verify_s": vs}}
    return out


# ---- CPU baselines (the oracle, timed on this host; reported, not the target) ------------------------------------------------------
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
