"""Summarise a rocprofv3 --pmc counter_collection csv: per kernel, mean of each counter over dispatches."""
import csv, glob, sys, collections
d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void mmf::", "").replace("mmf::", "")[:40]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(agg.items()):
    if not any(k in name for k in ("kernel",)):
        continue
    print(name, {k: round(sum(v) / len(v), 1) for k, v in sorted(cs.items())}, "n=%d" % len(next(iter(cs.values()))))
