"""profiles/traffic.json entry from the PMC pass directories of one workload (see tools/profile_r03.sh).
usage: pmc_to_traffic.py <instances> <f32|bf16> <source label> <dir> [<dir> ...]   -> prints the JSON object"""
import collections, csv, glob, json, sys
N, dtype, source = int(sys.argv[1]), sys.argv[2], sys.argv[3]
split = dtype == "f32x3"          # passes taken with MMF_GEMM=1: the bench's kernel trace calls those GEMM kernels *_split_kernel
if split:
    dtype = "f32"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[4:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void mmf::", "").replace("mmf::", "").split("<")[0]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"instances": N, "dtype": dtype, "source": source, "kernels": {}}
if split:
    out["gemm"] = "bf16x3"
for name, cs in sorted(agg.items()):
    if "kernel" not in name or "at::" in name:
        continue
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    e = {}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["FETCH_SIZE_KiB"] = round(m["FETCH_SIZE"], 1); e["WRITE_SIZE_KiB"] = round(m["WRITE_SIZE"], 1)
        e["bytes"] = int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
    for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"):
        if k in m:
            e[k] = round(m[k], 1)
    # the bench's kernel trace names both forms of the bf16 fused forward amil_fwd_fused_bf16_kernel and of K-dh dh_bf16_kernel
    alias = {"amil_fwd_fused2_bf16_kernel": "amil_fwd_fused_bf16_kernel", "dh2_bf16_kernel": "dh_bf16_kernel"}
    if split:
        alias = {"tn_kernel": "tn_split_kernel", "bwd_dh_kernel": "bwd_dh_split_kernel", "gate_fwd_mixed_kernel": "gate_fwd_split_kernel",
                 "gate_fwd_kernel": "gate_fwd_split_kernel", "linear_nt_split_kernel": "linear_nt_split_kernel"}
        if name not in alias:
            continue
    out["kernels"][alias.get(name, name)] = e
print(json.dumps(out, indent=1))
