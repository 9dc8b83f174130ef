"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv: short kernel names, calls, average us."""
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)):
    print("#", f)
    for r in csv.DictReader(open(f)):
        name = r["Name"].split("(")[0].replace("void ", "").replace("mmf::", "")
        if len(name) > 90: name = name[:87] + "..."
        print(f"{name:92s} calls {int(r['Calls']):5d}  avg_us {float(r['AverageNs'])/1e3:9.2f}  pct {float(r['Percentage']):5.1f}")
