#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into small text/JSON summaries that are
committed under profiles/.  Usage: summarise_profile.py <prof_dir> <tag>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d, tag = sys.argv[1], sys.argv[2]
out = {}
lines = []
stats = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    lines.append("== rocprofv3 --kernel-trace --stats : kernel_stats ==")
    for r in rows:
        lines.append(", ".join(f"{k}={r[k]}" for k in r))
    out["kernel_stats"] = rows
for name in ("fetch", "write"):
    f = glob.glob(os.path.join(d, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    agg = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        k = (r.get("Kernel_Name", "?").split("(")[0][:80], r.get("Counter_Name", "?"))
        agg[k][0] += float(r.get("Counter_Value", 0) or 0)
        agg[k][1] += 1
    lines.append(f"== rocprofv3 --pmc ({name}) : per-kernel mean counter value per dispatch ==")
    for (kn, cn), (s, c) in sorted(agg.items()):
        lines.append(f"{kn} {cn} mean={s / c:.6g} dispatches={c}")
        out.setdefault("pmc", {})[f"{kn}|{cn}"] = {"mean": s / c, "dispatches": c}
txt = "\n".join(lines)
print(txt)
os.makedirs(os.path.join(os.path.dirname(d), "summ"), exist_ok=True)
open(os.path.join(os.path.dirname(d), "summ", f"{tag}_summary.txt"), "w").write(txt + "\n")
json.dump(out, open(os.path.join(os.path.dirname(d), "summ", f"{tag}_summary.json"), "w"), indent=1)
