"""Builds libzkmle_amd.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libzkmle_amd.so")
ARCH = "gfx950"


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def deps():
    out = sources()
    out += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cuh", ".h"))]
    out.append(os.path.join(HERE, "..", "include", "zkmle.h"))
    return out


def build(force=False, verbose=False):
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps()):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, jobs = [], []
    for src in sources():
        obj = src[:-4] + ".o"
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps()):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
                   "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append(subprocess.Popen(cmd))      # one hipcc per translation unit, in parallel
        objs.append(obj)
    for j in jobs:
        if j.wait() != 0:
            raise subprocess.CalledProcessError(j.returncode, j.args)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
