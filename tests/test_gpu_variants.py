"""The environment switches that select another code path for the same result -- the transcript step on the device
(ZK_HOST_TRANSCRIPT=0), the rounds-per-pass cap of the basic sumcheck (ZK_BASIC_ROUNDS_PER_PASS, csrc/basic_multi.cuh) and the sparse GKR
prover's gate weights from a table instead of the eq half tables (ZK_GKR_WEIGHT_TABLE=1, csrc/zkmle_gkr_sparse.hip), one or two GKR rounds per
launch / exchange (ZK_GRID_TWO_ROUNDS, ZK_GRID_TWO_BITS, ZK_TAIL_TWO_ROUNDS: r4), the dense API on its dense tables (ZK_GKR_DENSE_TABLES=1) -- are read
once per process (the last one per call): each variant runs tests/_variant_worker.py in a child process and must reproduce the oracle's proofs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


DIGESTS = {}


@pytest.mark.parametrize("env", [{}, {"ZK_HOST_TRANSCRIPT": "0"}, {"ZK_BASIC_ROUNDS_PER_PASS": "1"}, {"ZK_BASIC_ROUNDS_PER_PASS": "2"},
                                 {"ZK_BASIC_ROUNDS_PER_PASS": "3"}, {"ZK_BASIC_ROUNDS_PER_PASS": "4"}, {"ZK_BASIC_ROUNDS_PER_PASS": "6"},
                                 {"ZK_BASIC_ROUNDS_PER_PASS": "8"}, {"ZK_GKR_WEIGHT_TABLE": "1"},
                                 {"ZK_GRID_TWO_ROUNDS": "0"}, {"ZK_TAIL_TWO_ROUNDS": "0"}, {"ZK_GRID_TWO_ROUNDS": "0", "ZK_TAIL_TWO_ROUNDS": "0"},
                                 {"ZK_GRID_TWO_BITS": "13"}, {"ZK_TAIL_TWO_ROUNDS": "16"}, {"ZK_GKR_DENSE_TABLES": "1"}],
                         ids=["default", "device_step", "k1", "k2", "k3", "k4", "k6", "k8", "weight_table",
                              "grid_single_rounds", "tail_single_rounds", "single_rounds", "grid_two_from_2p13", "tail_two_upto_16_pairs", "dense_tables"])
def test_variant_reproduces_oracle_proofs(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_variant_worker.py")], capture_output=True, text=True, env=e, timeout=600,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["checked"] >= 20
    assert out["mismatches"] == [], out
    # the sparse GKR proof of a 2^15-wide circuit: the same bytes whichever way the gate weights and the transcript step are computed
    DIGESTS[tuple(sorted(env.items()))] = out["sparse_gkr_digest"]
    assert out["sparse_gkr_digest"] == DIGESTS[()], (env, DIGESTS)


def test_host_side_stall_fails_the_proof_and_the_next_one_works():
    """Fault injection of the host-assisted transcript step (csrc/dev_transcript.cuh HostMailbox): the service thread of ONE proof is
    deaf for 3.5 s.  The kernels' spin budget runs out, every kernel still ends, the call returns an error that names the cause in
    well under 10 s, and the next proofs on the same thread -- basic and GKR -- equal the oracle's."""
    import time
    import numpy as np
    import __graft_entry__ as G
    from oracle import oracle as O
    zk = G.import_package()
    from zkmle_amd import _lib as L
    lib = zk.lib()
    L.check(lib.zk_init(0))
    if os.environ.get("ZK_HOST_TRANSCRIPT") == "0":
        pytest.skip("the transcript step runs on the device in this environment")
    os.environ.pop("ZK_ENABLE_FAULT_INJECTION", None)
    assert lib.zk_debug_stall_service_once(1) != 0              # the hook is process-global: it only arms where the environment asks for it
    os.environ["ZK_ENABLE_FAULT_INJECTION"] = "1"
    n = 1 << 15
    poly = zk.MultilinearPolynomial.random(0, n, 0xFA17)
    table = poly.evaluated_values
    want = O.sumcheck_basic_prove(0, table)
    assert np.array_equal(zk.Prover.init(0, poly).prove().round_univariate_polynomials, want[1])
    L.check(lib.zk_debug_stall_service_once(3500))
    t0 = time.time()
    with pytest.raises(zk.ZkError) as ei:
        zk.Prover.init(0, poly).prove()
    took = time.time() - t0
    assert took < 10.0, took
    assert "host" in str(ei.value).lower() or "host" in lib.zk_last_error().decode().lower(), str(ei.value)
    for _ in range(2):                                           # the thread's mailbox, service thread and arrival counters are usable again
        proof = zk.Prover.init(0, poly).prove()
        assert np.array_equal(proof.initial_claimed_sum, want[0]) and np.array_equal(proof.round_univariate_polynomials, want[1])
    tabs = np.stack([np.stack([zk.MultilinearPolynomial.random(0, 1 << 12, 0xFA20 + 2 * p + f).evaluated_values for f in range(2)]) for p in range(2)])
    claimed = O.vec_sum(0, O.sumpoly_reduce(0, tabs))
    co, ch = O.sumcheck_gkr_prove(0, tabs, claimed, O.Transcript())
    sp = zk.SumPolynomial([zk.ProductPolynomial([zk.MultilinearPolynomial(0, t) for t in prod]) for prod in tabs])
    L.check(lib.zk_debug_stall_service_once(3500))
    with pytest.raises(zk.ZkError):
        zk.sumcheck.prove(sp, claimed, zk.Transcript())
    res = zk.sumcheck.prove(sp, claimed, zk.Transcript())
    assert np.array_equal(res.round_univariate_polynomials, co) and np.array_equal(res.random_challenges, ch)
    os.environ.pop("ZK_ENABLE_FAULT_INJECTION", None)
