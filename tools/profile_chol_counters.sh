#!/bin/bash
# Run on the GPU box (via gpurun): counters of the ancestor-weight factorisation inside the information-form smoother
# (N_P=8192, m=512), three separate --pmc passes (matrix cores; FETCH_SIZE; WRITE_SIZE), no trace domains.
# Usage: tools/profile_chol_counters.sh <tag>
set -u
TAG=${1:-r01w}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="mag 8192 8 512 2 info"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o pmc -- python3 $REPO/tools/smoother_bench.py $ARGS > $OUT/mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $REPO/tools/smoother_bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $REPO/tools/smoother_bench.py $ARGS > $OUT/write.log 2>&1
cd $REPO
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
lines = []
for leg in ("pmc_mfma", "pmc_fetch", "pmc_write"):
    f = glob.glob(os.path.join(out, leg, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        lines.append(f"{leg}: no counter file"); continue
    agg = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        k = (r.get("Kernel_Name", "?").split("(")[0][:60], r.get("Counter_Name", "?"))
        agg[k][0] += float(r.get("Counter_Value", 0) or 0); agg[k][1] += 1
    lines.append(f"== {leg}: mean counter value per dispatch ==")
    for (kn, cn), (s, c) in sorted(agg.items()):
        if "chol_solve" in kn or "step_kernel" in kn:
            lines.append(f"{kn} {cn} mean={s / c:.6g} dispatches={c}")
os.makedirs(os.path.join(os.path.dirname(out), "summ"), exist_ok=True)
open(os.path.join(os.path.dirname(out), "summ", f"{tag}_chol_counters_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
