"""HBM traffic of one kernel from two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950;
MI355X_MICROARCH.md, HBM / rocprofv3 section): mean counter value per launch, FETCH_SIZE doubled (on gfx950 it reports exactly half of
the bytes of wide coalesced 16 B/lane reads), both in KiB.
    python3 tools/pmc_summary.py FETCH_DIR WRITE_DIR kernel-substring algorithmic_bytes > profiles/rN/<name>_pmc.json"""
import csv, glob, json, os, sqlite3, sys


def mean_counter(d, counter, kern):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and kern in r.get("Kernel_Name", ""):
                vals.append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        c = sqlite3.connect(f)
        try:
            cur = c.execute("select * from pmc_events limit 0")
            cols = [x[0] for x in cur.description]
            name_col = "counter_name" if "counter_name" in cols else ("name" if "name" in cols else None)
            kcol = "kernel_name" if "kernel_name" in cols else None
            vcol = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
            if name_col and vcol:
                q = f"select {vcol}" + (f", {kcol}" if kcol else "") + f" from pmc_events where {name_col} = ?"
                by_dispatch = {}
                for row in c.execute(q, (counter,)):
                    if kcol is None or kern in (row[1] or ""):
                        vals.append(float(row[0]))
        except sqlite3.Error:
            pass
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


fd, wd, kern, algo = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
fetch, nf = mean_counter(fd, "FETCH_SIZE", kern)
write, nw = mean_counter(wd, "WRITE_SIZE", kern)
out = {"kernel": kern, "FETCH_SIZE_KiB_mean_per_launch": fetch, "FETCH_SIZE_launches": nf, "WRITE_SIZE_KiB_mean_per_launch": write,
       "WRITE_SIZE_launches": nw, "algorithmic_bytes_per_launch": algo,
       "note": "FETCH_SIZE / WRITE_SIZE in KiB, separate rocprofv3 --pmc passes; FETCH_SIZE doubled per the gfx950 note (MI355X_MICROARCH.md, HBM section)"}
if fetch is not None and write is not None:
    out["hbm_read_bytes_corrected"] = 2 * fetch * 1024
    out["hbm_write_bytes"] = write * 1024
    out["hbm_bytes_per_launch"] = out["hbm_read_bytes_corrected"] + out["hbm_write_bytes"]
    out["traffic_over_algorithmic"] = out["hbm_bytes_per_launch"] / algo
print(json.dumps(out, indent=1))
