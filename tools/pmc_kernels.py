"""Per-kernel-name averages of the counters of one or more rocprofv3 --pmc passes (developer tool):
    python tools/pmc_kernels.py [--sha FINGERPRINT] [--note TEXT] <pass dir> [<pass dir> ...] > summary.json
With --sha the output is {"csrc_sha16": ..., "note": ..., "kernels": {...}} (bench.py reports `traffic` from it only when the
fingerprint matches the sources it runs); without it the plain {kernel: counters} map of round 2."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_aggregate import derived
argv = sys.argv[1:]
sha = note = None
while argv and argv[0] in ("--sha", "--note"):
    if argv[0] == "--sha": sha = argv[1]
    else: note = argv[1]
    argv = argv[2:]
agg = {}
for d in argv:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = (row["Dispatch_Id"], row["Kernel_Name"].split("(")[0][:100])
                per.setdefault(k, {})
                per[k][row["Counter_Name"]] = per[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        for (_, name), ctr in per.items():
            e = agg.setdefault(name, {})
            for c, v in ctr.items():
                s = e.setdefault(c, [0.0, 0])
                s[0] += v; s[1] += 1
out = {}
for name, e in sorted(agg.items()):
    o = {c: s[0] / s[1] for c, s in e.items()}
    o["dispatches"] = max(s[1] for s in e.values())
    derived(o)
    if "FETCH_SIZE" in o:
        o["fetch_bytes_corrected"] = 2.0 * o["FETCH_SIZE"] * 1024.0
    if "WRITE_SIZE" in o:
        o["write_bytes"] = o["WRITE_SIZE"] * 1024.0
        o["hbm_bytes_per_launch"] = o.get("fetch_bytes_corrected", 0.0) + o["write_bytes"]
    out[name] = o
print(json.dumps({"csrc_sha16": sha, "note": note, "kernels": out} if sha else out, indent=1))
