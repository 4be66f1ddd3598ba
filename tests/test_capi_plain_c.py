"""The boundary is plain C: include/zkmle.h compiles as C99 with gcc -Wall -Werror -pedantic, and a C program linked
only against libzkmle_amd.so drives the path (tools/capi_example.c).  On a GPU it reproduces the reference's first
known answers; without one it observes the loud ZK_E_NO_DEVICE."""
import os
import subprocess

import pytest

import __graft_entry__ as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_and_run(tmp_path):
    G.build()
    libdir = os.path.join(ROOT, "zk-cryptography-research-implementations_amd")
    exe = str(tmp_path / "capi_example")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", os.path.join(ROOT, "tools", "capi_example.c"), "-o", exe,
                           "-L" + libdir, "-lzkmle_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return subprocess.run([exe], capture_output=True, text=True, timeout=120)


def test_header_is_c99_and_library_links_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by the gpu variant")
    out = _build_and_run(tmp_path)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "no HIP device" in out.stdout and "-9" in out.stdout


@pytest.mark.gpu
def test_plain_c_program_reproduces_reference_kats(tmp_path):
    out = _build_and_run(tmp_path)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "[18, 48]" in out.stdout and "-> 78" in out.stdout and "power of 2" in out.stdout
