"""One step's kernel timeline out of a rocprofv3 --kernel-trace csv directory: start / end (µs from the step's first kernel),
queue, name.  usage: timeline.py DIR [marker-kernel-substring] [step-index-from-the-end]
A step is delimited by successive launches of the marker kernel (default: the first kernel name that occurs once per step)."""
import csv, glob, os, sys
d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "linear_nt_kernel<mmf::Tile<208"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(idx) < back + 2:
    print("marker not found often enough:", len(idx)); sys.exit(1)
# a step: from some kernels before the marker (the small branches start first) -- take the window between the END of the previous
# step's last kernel gap: simply print from marker[-back-1] to marker[-back] shifted by the kernels in front
a, b = idx[-back - 1], idx[-back]
t0 = int(rows[a]["Start_Timestamp"])
qs = {}
for r in rows[a - 6:b - 6 if b - 6 > a else b]:
    q = qs.setdefault(r["Queue_Id"], len(qs))
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:7.1f}  q{q}  {'    ' * q}{r['Kernel_Name'][:90]}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))}")
