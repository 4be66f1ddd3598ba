"""GPU: the small programs built from the library's own kernel headers that check device code no C-ABI call isolates.
tools/test_quad_ops.bin: the four-lane point operations of csrc/g1u.cuh against the one-lane ones, and the bucket reduction's weighted sums by
bits (csrc/msm_bits.cuh) on arrays of known multiples of the generator -- including arrays with points at infinity in some quads of a wave and
not in others, the case in which a branch in front of the cross-lane moves gave wrong doublings (r4)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_quad_point_operations_and_weighted_sums_by_bits():
    exe = os.path.join(ROOT, "tools", "test_quad_ops.bin")
    assert os.path.exists(exe), "tools/test_quad_ops.bin is built by __graft_entry__.build()"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]
    assert "mismatch mask 0x0" in p.stdout and p.stdout.count(": ok") == 32 and "DIFFERS" not in p.stdout


@pytest.mark.parametrize("args", [["16", "3", "256"], ["18", "3", "1536", "3"]])
def test_round_kernels_alone_against_plain_kernels(args):
    """tools/microbench_round.bin: the two GKR round kernels (csrc/sumcheck_kernels.cuh: uniform-multiplier fold, lazy sums, nodes 0 / 1 / infinity) launched on
    their own, the folded tables byte for byte and the sums mod p against plain kernels written in the tool (general field product, one modular operation
    after the other) -- on 4 tables and on 3 tables + a constant factor, with the multiplier read and worked out per wave"""
    import json
    exe = os.path.join(ROOT, "tools", "microbench_round.bin")
    assert os.path.exists(exe), "tools/microbench_round.bin is built by __graft_entry__.build()"
    p = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]
    checked = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{") and "matches_reference" in ln]
    assert len(checked) == 6 and all(c["matches_reference"] for c in checked), p.stdout[-3000:]
