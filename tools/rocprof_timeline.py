"""Dispatch timeline of the LAST `n` kernels of a rocprofv3 --kernel-trace run (rocpd SQLite `*_results.db`): start, end, duration (us)
relative to the first of them.  With a kernel-name substring as third argument the window starts at the last dispatch of that kernel.
    python3 tools/rocprof_timeline.py DIR 12 [first-kernel-substring]"""
import glob, os, sqlite3, sys

d, n = sys.argv[1], int(sys.argv[2])
first = sys.argv[3] if len(sys.argv) > 3 else None
for f in sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)):
    rows = list(sqlite3.connect(f).execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
    if first:
        idx = max(i for i, r in enumerate(rows) if first in r[0])
        rows = rows[idx:idx + n]
    else:
        rows = rows[-n:]
    t0 = rows[0][1]
    for name, s, e, g, w in rows:
        print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  grid {g:>8} wg {w:>5}  {name[:110]}")
    print(f"# span {(rows[-1][2] - t0) / 1e3:.1f} us, kernel time {sum(e - s for _, s, e, _, _ in rows) / 1e3:.1f} us")
