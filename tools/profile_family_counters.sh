#!/bin/bash
# Run on the GPU box (via gpurun): counters of the family-products form of the filter's read-only steps (rbpf_options.family_products = 1,
# N = 65536, m = 512, lazy_depth 4): three separate --pmc passes (matrix cores; FETCH_SIZE; WRITE_SIZE), no trace domains.
# Usage: tools/profile_family_counters.sh <tag>
set -u
TAG=${1:-r04}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_family_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--lazy 4 --family 1 --steps 12 --warmup 5"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o pmc -- python3 $REPO/tools/family_ab.py $ARGS > $OUT/mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $REPO/tools/family_ab.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $REPO/tools/family_ab.py $ARGS > $OUT/write.log 2>&1
cd $REPO
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
lines = [f"== rocprofv3 --pmc (three separate passes): tools/family_ab.py --lazy 4 --family 1 --steps 12 --warmup 5 (N = 65536, m = 512) =="]
for leg in ("pmc_mfma", "pmc_fetch", "pmc_write"):
    f = glob.glob(os.path.join(out, leg, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        lines.append(f"{leg}: no counter file"); continue
    agg = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        k = (r.get("Kernel_Name", "?").split("(")[0][:70], r.get("Counter_Name", "?"))
        agg[k][0] += float(r.get("Counter_Value", 0) or 0); agg[k][1] += 1
    lines.append(f"-- {leg}: mean counter value per dispatch --")
    for (kn, cn), (s, c) in sorted(agg.items()):
        if "family" in kn or "step_sym" in kn:
            lines.append(f"{kn:72s} {cn:28s} mean={s / c:.6g} dispatches={c}")
lines.append("(FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE reports half of wide coalesced reads: bytes = 2 * FETCH * 1024)")
open(os.path.join(os.path.dirname(out), f"{tag}_family_counters_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
