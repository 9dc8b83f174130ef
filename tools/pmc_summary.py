"""Summarise rocprofv3 --pmc counter_collection csvs: per kernel, mean of each counter over dispatches.
usage: pmc_summary.py <dir> [<dir> ...]   (one directory per --pmc pass; counters are merged per kernel)"""
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void mmf::", "").replace("mmf::", "").split("<")[0][:40]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(agg.items()):
    if "kernel" not in name:
        continue
    m = {k: round(sum(v) / len(v), 1) for k, v in sorted(cs.items())}
    extra = ""
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:      # KiB; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section)
        extra = "  HBM bytes/launch = %.1f MB" % ((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024 / 1e6)
    print(name, m, "n=%d" % len(next(iter(cs.values()))), extra)
