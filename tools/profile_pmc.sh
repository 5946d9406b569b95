#!/bin/bash
# Run on the GPU box (via gpurun): HBM counters of selected kernels of one tools/smoother_bench.py run, two separate
# `rocprofv3 --pmc` passes (FETCH_SIZE; WRITE_SIZE) as MI355X_MICROARCH.md prescribes (no trace domains with --pmc).
# Usage: tools/profile_pmc.sh <tag> <kernel-name-substrings,comma-separated> <smoother_bench.py args...>
set -u
TAG=$1; FILTER=$2; shift 2
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o pmc -- python3 $REPO/tools/smoother_bench.py "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o pmc -- python3 $REPO/tools/smoother_bench.py "$@" > $OUT/write.log 2>&1
cd $REPO
python3 - "$OUT" "$TAG" "$FILTER" "$*" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, tag, filt, argstr = sys.argv[1:5]
keys = [k for k in filt.split(",") if k]
lines = [f"== rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes): tools/smoother_bench.py {argstr} ==",
         "counter unit KiB; on gfx950 FETCH_SIZE reports half of wide coalesced reads (MI355X_MICROARCH.md): bytes = 2 * FETCH * 1024"]
tot = defaultdict(dict)
for leg in ("fetch", "write"):
    f = glob.glob(os.path.join(out, leg, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        lines.append(f"{leg}: no counter file"); continue
    agg = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        kn = r.get("Kernel_Name", "?").split("(")[0]
        if keys and not any(k in kn for k in keys):
            continue
        agg[(kn[:90], r.get("Counter_Name", "?"))][0] += float(r.get("Counter_Value", 0) or 0)
        agg[(kn[:90], r.get("Counter_Name", "?"))][1] += 1
    for (kn, cn), (s, c) in sorted(agg.items()):
        mean = s / c
        gb = mean * 1024 * (2 if cn == "FETCH_SIZE" else 1) / 1e9
        lines.append(f"{kn:90s} {cn:10s} mean={mean:12.6g} KiB  dispatches={c:5d}  -> {gb:8.3f} GB per dispatch")
        tot[kn][cn] = gb
for kn, d in tot.items():
    if len(d) == 2:
        lines.append(f"{kn:90s} HBM traffic per dispatch = {d['FETCH_SIZE'] + d['WRITE_SIZE']:.3f} GB")
os.makedirs(os.path.join(os.path.dirname(out), "summ"), exist_ok=True)
open(os.path.join(os.path.dirname(out), "summ", f"{tag}_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
