"""CPU: the product's unsaturated (29-bit limb) Montgomery arithmetic against its saturated CIOS reference form,
all four fields (tools/ufield_selftest.hip, compiled with hipcc and run on the host)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_unsaturated_arithmetic_selftest(tmp_path):
    exe = str(tmp_path / "ufield_selftest")
    src = os.path.join(ROOT, "tools", "ufield_selftest.hip")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O2", "-std=c++17", src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    for name in ("Fr381", "Fq381", "Bn254Fq", "Bn254Fr"):
        assert f"{name}: ok" in out.stdout
