#!/bin/bash
# HBM-side counters of the two factorisation kernels on their own (tools/chol_bench.py, covariance-form mode: no
# Imat gather), separate --pmc passes.  Usage: tools/profile_chol_bench_counters.sh <tag> [M] [batch]
set -u
TAG=${1:-r01w}; M=${2:-515}; B=${3:-2048}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_${TAG}_cb
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- python3 $REPO/tools/chol_bench.py --M $M --batch $B --reps 2 > $OUT/$C.log 2>&1
done
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for leg in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE"):
    f = glob.glob(os.path.join(out, leg, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        print(leg, "no counter file"); continue
    agg = defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        agg[(r.get("Kernel_Name", "?").split("(")[0][:50], r.get("Counter_Name", "?"))].append(float(r.get("Counter_Value", 0) or 0))
    for (kn, cn), v in sorted(agg.items()):
        if "chol_solve" in kn:
            print(kn, cn, "max per dispatch =", max(v), "dispatches", len(v))
PY
