#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs written by scripts/prof_k1.sh: per-kernel mean of every
counter + the kernel-trace stats, as one small text file fit for profiles/."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
pats = (sys.argv[2] if len(sys.argv) > 2 else "k1_").split("|")   # kernel name substrings, any of them
out = []
st = os.path.join(root, "trace", "k1_kernel_stats.csv")
if os.path.exists(st):
    out.append("== kernel-trace --stats (ns) ==")
    for r in csv.DictReader(open(st)):
        name = r["Name"][:70]
        if not any(p_ in r["Name"] for p_ in pats):
            continue
        out.append(f'{name:70s} calls={r["Calls"]:>4s} avg_ns={float(r["AverageNs"]):12.0f} '
                   f'min={r["MinNs"]:>10s} max={r["MaxNs"]:>10s} pct={r["Percentage"]}')
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    f = os.path.join(d, "k1_counter_collection.csv")
    if not os.path.exists(f):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(f)):
        if any(p_ in r["Kernel_Name"] for p_ in pats):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])
    for k in acc:
        out.append(f"== {os.path.basename(d)} :: {k}  grid={meta[k][0]} wg={meta[k][1]} lds={meta[k][2]} vgpr={meta[k][3]} sgpr={meta[k][4]} ==")
        for c, v in sorted(acc[k].items()):
            out.append(f"  {c:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
print("\n".join(out))
