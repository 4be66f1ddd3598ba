"""Summarise a rocprofv3 --kernel-trace --stats output directory (rocpd SQLite `*_results.db`, or the CSV files of
--output-format csv): per-kernel calls / total / average / min / max, the kernel-busy time against the traced span, and for the
kernels named on the command line the dispatches grouped by grid size.
    python3 tools/rocprof_summary.py DIR [kernel-substring ...]"""
import csv, glob, os, sqlite3, sys

d = sys.argv[1]
want = sys.argv[2:]


def report(rows):                       # rows: (name, start_ns, end_ns, grid_x, wg_x)
    by = {}
    for n, s, e, g, w in rows:
        by.setdefault(n, []).append((e - s) / 1e3)
    tot = sum(sum(v) for v in by.values())
    span = (max(r[2] for r in rows) - min(r[1] for r in rows)) / 1e3 if rows else 0
    print(f"{'kernel':90} {'calls':>6} {'total_ms':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'%':>6}")
    for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:45]:
        print(f"{n[:90]:90} {len(v):>6} {sum(v) / 1e3:>10.3f} {sum(v) / len(v):>9.1f} {min(v):>9.1f} {max(v):>9.1f} {100 * sum(v) / tot:>6.2f}")
    print(f"# kernel time {tot / 1e3:.3f} ms over a traced span of {span / 1e3:.3f} ms ({len(rows)} dispatches)")
    for w_ in want:
        sel = [r for r in rows if w_ in r[0]]
        grids = {}
        for n, s, e, g, w in sel:
            grids.setdefault((g, w), []).append((e - s) / 1e3)
        if sel:
            print(f"# {w_}: dispatches by grid size (us)")
        for (g, w), v in sorted(grids.items(), key=lambda kv: -max(kv[1]))[:10]:
            v = sorted(v)
            print(f"   grid {g:>10} wg {w:>5}  n={len(v):<4} median {v[len(v) // 2]:9.1f}  min {v[0]:9.1f}  max {v[-1]:9.1f}")


for f in sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)):
    print("#", os.path.relpath(f, d))
    c = sqlite3.connect(f)
    report(list(c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start")))
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
    print("#", os.path.relpath(f, d))
    rows = list(csv.DictReader(open(f)))
    report([(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0),
             int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 0)) for r in rows])
