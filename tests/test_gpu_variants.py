"""The environment switches that select another code path for the same result -- the transcript step on the device
(ZK_HOST_TRANSCRIPT=0), the rounds-per-pass cap of the basic sumcheck (ZK_BASIC_ROUNDS_PER_PASS, csrc/basic_multi.cuh) and the sparse GKR
prover's gate weights from a table instead of the eq half tables (ZK_GKR_WEIGHT_TABLE=1, csrc/zkmle_gkr_sparse.hip) -- are read
once per process: each variant runs tests/_variant_worker.py in a child process and must reproduce the oracle's proofs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


DIGESTS = {}


@pytest.mark.parametrize("env", [{}, {"ZK_HOST_TRANSCRIPT": "0"}, {"ZK_BASIC_ROUNDS_PER_PASS": "1"}, {"ZK_BASIC_ROUNDS_PER_PASS": "2"},
                                 {"ZK_BASIC_ROUNDS_PER_PASS": "3"}, {"ZK_GKR_WEIGHT_TABLE": "1"}],
                         ids=["default", "device_step", "k1", "k2", "k3", "weight_table"])
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
