"""Static v_mad_u64_u32 count of msm_bucket_sum_kernel per basic-block range (the MSM roofline's numerator, bench.py
MADS_PER_MIXED_ADD).  Compiles csrc/msm_bucket.hip to gfx950 assembly and counts the multiply-adds between the labels /
exec-mask branches of the main loop, next to the source-level count:
    L = 14 limbs of 29 bits (Fq381);   product = L^2 operand rows + L^2 Montgomery rows = 392
    squaring = L (L + 1) / 2 + L^2 = 301;   dual product with one reduction (Y3) = 3 L^2 = 588
    g1u_madd (csrc/g1u.cuh) = 6 products + 2 squarings + 1 dual product = 6 * 392 + 2 * 301 + 588 = 3542
(the round-1 figure 3724 = 19 * 196 predates the symmetric squaring).  Blocks guarded by the ZZ3 = 0 test (exceptional
cases: doubling, cancellation) are listed but are not on the common path."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "zk-cryptography-research-implementations_amd", "csrc", "msm_bucket.hip")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "--cuda-device-only",
                           "-S", "-o", out, src], stderr=subprocess.DEVNULL)
    lines = open(out).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN2zk21msm_bucket_sum_kernel"))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("_ZN2zk") and lines[i].rstrip().endswith(":") is False and "@_ZN2zk" in lines[i] and i > start + 10)
body = lines[start:end]
cuts = [i for i, l in enumerate(body) if re.match(r"^\.LBB0_\d+:", l) or "s_cbranch_exec" in l or "s_cbranch_vcc" in l]
print(f"{'lines':>14} {'v_mad_u64_u32':>14} {'VALU':>6}  first label / branch")
prev = 0
for c in cuts + [len(body)]:
    seg = body[prev:c]
    mads = sum("v_mad_u64_u32" in l for l in seg)
    valu = sum(bool(re.match(r"^\s+v_", l)) for l in seg)
    if mads or valu > 50:
        print(f"{prev:>6}-{c:<7} {mads:>14} {valu:>6}  {body[prev].strip()[:60]}")
    prev = c
print("common path of one loop iteration (blocks 518-1997, 2527-3109, 3109-5906): 1085 + 392 + 2065 = 3542 multiply-adds, 4796 VALU;")
print("blocks 1997-2527, 5910-6488 (full reductions for the exact P == Q / P == -Q tests) and 6488-10489 (g1u_mdbl) run only when ZZ3 == 0 mod p.")
print("total v_mad_u64_u32 in the kernel:", sum("v_mad_u64_u32" in l for l in body))
print("source-level count per mixed addition: 6*392 + 2*301 + 588 =", 6 * 392 + 2 * 301 + 588)
