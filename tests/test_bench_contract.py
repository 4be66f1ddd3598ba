"""bench.py's contract with the driver: the rank count it reports is the rank count it ran (CPU part), and on a GPU the JSON line has
the keys the driver and the judge read (one small run and one two-rank rehearsal through the self-launcher)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env=None, timeout=600):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=e, timeout=timeout, cwd=ROOT)


def test_world_size_must_match_gpus_flag():
    # under torchrun bench.py is ONE rank: a line for the wrong rank count must not be printed
    p = run(["--gpus", "1", "--steps", "1"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=2 but --gpus 1" in (p.stderr + p.stdout)
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_rank_count_must_be_a_power_of_two():
    p = run(["--gpus", "3", "--steps", "1"], env={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "power of two" in (p.stderr + p.stdout)


def last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_one_gpu_line_has_the_contract_keys():
    p = run(["--steps", "5", "--warmup", "2", "--log-n", "16", "--msm-reps", "1", "--no-configs"])
    assert p.returncode == 0, p.stderr[-2000:]
    d = last_json(p.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "prewarm_s", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "collectives", "post_check"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
    assert d["post_check"]["bit_exact_vs_oracle"] is True
    m = d["msm"]
    assert m["roofline"]["bound"].startswith("valu") and 0 < m["roofline"]["frac"] < 1.2 and m["cpu_baseline"]["all_cores"]["cores"] >= 1
    assert m["post_check"]["holds"] is True and m["terms_per_s"] > 0 and m["window_bits"] >= 1
    c5 = d["config5_strong"]
    assert c5["sumcheck"]["verifier_equations_hold"] is True and c5["msm"]["same_point_on_every_rank"] is True
    assert c5["msm"]["post_check"]["holds"] is True


@pytest.mark.gpu
def test_one_gpu_line_times_every_named_path_and_config():
    # the functions north_star names beyond the fold and the MSM (evaluate, setup, commit, open_and_prove, dense gkr::prove) and BASELINE configs 1-4,
    # each timed AND checked; sizes follow --log-n where they are not fixed by BASELINE
    p = run(["--steps", "5", "--warmup", "2", "--log-n", "16", "--msm-reps", "1", "--no-msm", "--no-config5"])
    assert p.returncode == 0, p.stderr[-2000:]
    d = last_json(p.stdout)
    pa, cf = d["paths"], d["configs"]
    assert pa["evaluate_2p16"]["post_check"]["equals_chain_of_partial_evaluate"] is True and pa["evaluate_2p16"]["ms"] > 0
    assert pa["kzg_open_2p16"]["post_check"]["pairing_check_of_the_timed_opening"] is True and pa["kzg_open_2p16"]["post_check"]["pairings"] == 17
    assert pa["setup_2p16"]["s"] > 0 and pa["kzg_commit_2p16"]["terms_per_s"] > 0
    assert pa["gkr_dense"]["post_check"]["verifier_accepts"] is True and pa["gkr_dense"]["post_check"]["both_representations_same_bytes"] is True
    assert pa["gkr_dense"]["dense_tables_ms"] > 0
    assert pa["gkr_succinct"]["ms"] > 0 and pa["gkr_succinct"]["post_check"]["verify_succinct_accepts"] is True
    assert cf["cfg1"]["post_check"]["gpu_proof_equals_oracle_proof"] is True and cf["cfg1"]["cpu"]["cores"] == 1
    assert cf["cfg2"]["absorb_GBps"] > 0 and cf["cfg4"]["absorb_GBps"] > 0
    rk = cf["cfg4"]["round_kernels"]
    assert 0 < rk["fold_round_evals_kernel"]["frac_valu"] < 1 and 0 < rk["round_evals_kernel"]["frac_valu"] < 1


@pytest.mark.gpu
def test_eight_rank_rehearsal_is_strong_scaling_of_one_table():
    # BASELINE config 5's split on one GPU: 8 ranks as threads of the bench process (a one-GPU box allows 6 processes on its card)
    p = run(["--gpus", "8", "--rehearse", "--steps", "4", "--warmup", "1", "--log-n", "16", "--msm-reps", "1", "--no-cpu-baseline"])
    assert p.returncode == 0, p.stderr[-2000:]
    d = last_json(p.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["entries_per_gpu"] == (1 << 16) // 8
    assert d["collectives"]["library"] == "local-threads" and "rehearsal" in d["collectives"]["backend"]
    assert "failed_legs" not in d
    c5 = d["config5_strong"]
    assert c5["sumcheck"]["verifier_equations_hold"] is True and c5["sumcheck"]["rounds"] == 16
    assert c5["msm"]["same_point_on_every_rank"] is True and c5["msm"]["post_check"]["holds"] is True
    assert "ms_per_proof" in d["sharded_sumcheck"] and d["weak"]["value"] > 0


@pytest.mark.gpu
def test_preflight_exercises_every_collective_and_one_proof_of_each_kind():
    # what the driver can run first on a multi-GPU node (seconds): here the ranks are threads of one process on the one GPU
    p = run(["--gpus", "4", "--rehearse", "--preflight"])
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    d = last_json(p.stdout)
    assert d["preflight"] is True and d["ok"] is True and d["n_gpus"] == 4 and d["seconds"] < 120
    for k in ("all_reduce_i64_27_words_exact", "all_reduce_i64_1152_words_exact", "all_gather_rank_order", "broadcast_208_from_rank_0", "broadcast_208_from_rank_3",
              "sharded_basic_sumcheck_with_absorb_equals_oracle", "sharded_evaluate_equals_oracle", "sharded_gkr_sumcheck_equals_oracle", "sharded_msm_identity",
              "every_rank_agrees"):
        assert d["checks"][k] is True, k
    one = run(["--gpus", "1", "--preflight"])                # the same at one rank (no transport at all)
    assert one.returncode == 0 and last_json(one.stdout)["ok"] is True


@pytest.mark.gpu
def test_config5_at_its_stated_size_eight_ranks_one_2p24_table():
    # BASELINE config 5 as stated -- ONE 2^24-entry table and ONE 2^24-term MSM split 8 ways -- inside the suite: the 8 ranks are threads sharing the
    # one GPU (not a scaling measurement); what the line checks must hold: the verifier's equations on the sharded proof, the same MSM point on every
    # rank, the O(N) identity of the 2^24 MSM
    p = run(["--gpus", "8", "--rehearse", "--log-n", "24", "--steps", "20", "--warmup", "5", "--msm-reps", "1", "--no-cpu-baseline", "--no-weak"])
    assert p.returncode == 0, p.stderr[-2000:]
    d = last_json(p.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["entries_per_gpu"] == (1 << 24) // 8 and "failed_legs" not in d
    c5 = d["config5_strong"]
    assert c5["sumcheck"]["verifier_equations_hold"] is True and c5["sumcheck"]["rounds"] == 24
    assert c5["msm"]["same_point_on_every_rank"] is True and c5["msm"]["post_check"]["holds"] is True
    assert c5["msm_precomputed"]["same_point_as_plain"] is True


@pytest.mark.gpu
def test_a_stuck_secondary_leg_does_not_take_the_headline_with_it():
    # N > 1: the legs after the headline run collectives; past ZK_BENCH_LEG_DEADLINE the line is printed with what has finished (bench.py
    # secondary_legs_watchdog).  Here the deadline is shorter than the legs take, on the 8-rank rehearsal.
    p = run(["--gpus", "8", "--rehearse", "--steps", "4", "--warmup", "1", "--log-n", "18", "--msm-reps", "1", "--no-cpu-baseline"],
            env={"ZK_BENCH_LEG_DEADLINE": "0.05"})
    assert p.returncode == 0, p.stderr[-2000:]
    d = last_json(p.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["value"] > 0 and "roofline" in d
    assert any("watchdog" in leg for leg in d["failed_legs"])
    p = run(["--gpus", "8", "--rehearse", "--steps", "4", "--warmup", "1", "--log-n", "18", "--msm-reps", "1", "--no-cpu-baseline", "--require-rccl"],
            env={"ZK_BENCH_LEG_DEADLINE": "0.05"})
    assert p.returncode != 0
