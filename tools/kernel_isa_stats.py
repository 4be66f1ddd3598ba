"""Static instruction mix and register use of the kernels of one translation unit (gfx950 assembly from hipcc -S):
    python tools/kernel_isa_stats.py csrc/zkmle_sumcheck.hip | file.s [substring of the demangled kernel name ...]
Per kernel: VGPRs, SGPRs, scratch, and -- for the whole kernel and for its largest loop body (the basic blocks between the first
backward branch target and that branch) -- the number of v_mad_u64_u32, other VALU, SALU, vector-memory and LDS instructions.
A static count: every basic block is counted once, so it equals the dynamic per-iteration count only for straight-line loop bodies
(the round kernels' loops are; the rare-path blocks -- fe_from_u_below_2p's second subtraction -- are listed as `cold`)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def classify(line):
    m = re.match(r"^\s+([a-z_0-9]+)", line)
    if not m:
        return None
    op = m.group(1)
    if op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64_i32"):
        return "mad64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def count(lines):
    c = {"mad64": 0, "valu": 0, "salu": 0, "vmem": 0, "lds": 0, "other": 0}
    for l in lines:
        k = classify(l)
        if k:
            c[k] += 1
    return c


def main():
    src = sys.argv[1]
    if not os.path.isabs(src):
        cand = os.path.join(ROOT, "zk-cryptography-research-implementations_amd", src)
        src = cand if os.path.exists(cand) else os.path.join(ROOT, src)
    pats = sys.argv[2:]
    if src.endswith(".s"):                                  # an assembly file kept from an earlier run
        text = open(src).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "--cuda-device-only",
                                   "-S", "-o", out, src], stderr=subprocess.DEVNULL)
            text = open(out).read()
    lines = text.splitlines()
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+):", l)] if m]
    names = demangle([n for _, n in starts])
    meta = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        body = m.group(2)
        g = lambda key: (re.search(r"\.amdhsa_%s (\S+)" % key, body) or [None, "?"])[1]
        meta[m.group(1)] = {"vgpr": g("next_free_vgpr"), "sgpr": g("next_free_sgpr"), "scratch": g("private_segment_fixed_size"), "accum_offset": g("accum_offset")}
    for k, (i, n) in enumerate(starts):
        dn = names[n]
        if pats and not any(p in dn for p in pats):
            continue
        end = next((j for j in range(i + 1, len(lines)) if lines[j].startswith(".Lfunc_end")), len(lines))
        body = lines[i:end]
        labels = {m.group(1): j for j, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for j, l in enumerate(body):
            m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"^\s+s_branch\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < j:
                loops.append((labels[m.group(1)], j))
        tot = count(body)
        print(f"== {dn[:150]}")
        print(f"   {meta.get(n, {})}")
        print(f"   whole kernel: {tot}")
        if loops:
            a, b = max(loops, key=lambda ab: ab[1] - ab[0])
            print(f"   largest loop (lines {a}-{b}): {count(body[a:b])}")
            # cold blocks inside the loop: blocks entered only through a forward s_cbranch_execz skip
            cold, j = [], a
            while j < b:
                m = re.match(r"^\s+s_cbranch_execz\s+(\.LBB\d+_\d+)", body[j])
                if m and m.group(1) in labels and labels[m.group(1)] > j and labels[m.group(1)] <= b:
                    cold.append((j, labels[m.group(1)]))
                j += 1
            for ca, cb in cold:
                print(f"     block skipped when no lane needs it (lines {ca}-{cb}): {count(body[ca:cb])}")


main()
