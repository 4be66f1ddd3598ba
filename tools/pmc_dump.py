"""Mean per-launch value of every counter of a `rocprofv3 --pmc ...` run (rocpd SQLite `*_results.db`), per kernel whose name holds
one of the given substrings.   python3 tools/pmc_dump.py DIR substring [substring ...]"""
import collections, glob, os, sqlite3, sys

d, subs = sys.argv[1], sys.argv[2:]
for f in sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)):
    c = sqlite3.connect(f)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table', 'view')")]
    if "counters_collection" in tabs:
        cols = [x[1] for x in c.execute("pragma table_info(counters_collection)")]
        q = "select kernel_name, counter_name, value, dispatch_id from counters_collection" if "kernel_name" in cols else None
    else:
        q = None
    if q is None:
        print("# tables:", tabs)
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))      # (kernel, dispatch) -> counter -> sum over instances
    for kn, cn, v, did in c.execute(q):
        for s in subs:
            if s in kn:
                acc[(s, did)][cn] += float(v)
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for (s, did), cs in acc.items():
        for cn, v in cs.items():
            per[s][cn].append(v)
    for s in subs:
        print(f"## {s}: {max((len(v) for v in per[s].values()), default=0)} launches")
        for cn in sorted(per[s]):
            v = per[s][cn]
            print(f"  {cn:28s} mean {sum(v) / len(v):16.1f}   max {max(v):16.1f}")
