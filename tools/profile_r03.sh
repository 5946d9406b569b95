#!/bin/bash
# r03 profiles of the information-form smoother at the per-GPU share (N_P = 8192, m = 512, symmetric covariance storage), run on the
# GPU box via gpurun: rocprofv3 kernel stats of the default and the carried-factor configuration, and matrix-core counters of the exact
# factorisation (separate --pmc passes, no trace domains with --pmc).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03_prof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -o sm -- python3 $REPO/tools/smoother_bench.py mag 8192 120 512 2 info lazy_depth=3 storage=2 > $OUT/default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/carried -o sm -- python3 $REPO/tools/smoother_bench.py mag 8192 120 512 2 info lazy_depth=3 chol_refresh=32 storage=2 > $OUT/carried.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 --output-format csv -d $OUT/mfma -o pmc -- python3 $REPO/tools/smoother_bench.py mag 8192 24 512 2 info lazy_depth=3 storage=2 > $OUT/mfma.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU --output-format csv -d $OUT/grbm -o pmc -- python3 $REPO/tools/smoother_bench.py mag 8192 24 512 2 info lazy_depth=3 storage=2 > $OUT/grbm.log 2>&1
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
lines = []
for leg, title in (("default", "lazy_depth 3, fresh factorisation every step"), ("carried", "lazy_depth 3, chol_refresh 32")):
    f = glob.glob(os.path.join(out, leg, "**", "*kernel_stats.csv"), recursive=True)
    lines.append(f"== rocprofv3 --kernel-trace --stats: tools/smoother_bench.py mag 8192 120 512 2 info storage=2 ({title}) ==")
    if not f:
        lines.append("no stats file"); continue
    for r in list(csv.DictReader(open(f[0])))[:12]:
        lines.append(f"{r['Name'][:96]:96s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e6:9.3f} ms  total {float(r['TotalDurationNs'])/1e6:10.2f} ms  {float(r['Percentage']):6.2f} %")
lines.append("== rocprofv3 --pmc (separate passes), chol_solve64_kernel<1, 8>, means per dispatch ==")
agg = defaultdict(lambda: [0.0, 0])
for leg in ("mfma", "grbm"):
    for f in glob.glob(os.path.join(out, leg, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "chol_solve64" in r.get("Kernel_Name", ""):
                a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, c) in sorted(agg.items()):
    lines.append(f"{k:28s} mean {s / c:14.6g}  dispatches {c}")
if "SQ_VALU_MFMA_BUSY_CYCLES" in agg and "GRBM_GUI_ACTIVE" in agg:
    busy = agg["SQ_VALU_MFMA_BUSY_CYCLES"][0] / agg["SQ_VALU_MFMA_BUSY_CYCLES"][1]
    gui = agg["GRBM_GUI_ACTIVE"][0] / agg["GRBM_GUI_ACTIVE"][1]
    lines.append(f"matrix pipes busy: {busy:.4g} / ({gui:.4g} / 8 XCDs x 1024 SIMDs) = {busy / (gui / 8 * 1024) * 100:.1f} % of the SIMD-cycles")
if "SQ_INSTS_VALU_MFMA_F64" in agg:
    n = agg["SQ_INSTS_VALU_MFMA_F64"][0] / agg["SQ_INSTS_VALU_MFMA_F64"][1]
    lines.append(f"fp64 MFMA instructions {n:.4g} x 2048 flop = {n * 2048 / 1e9:.1f} GFLOP issued per launch (373 GFLOP algorithmic at n = 515, 8192 matrices)")
os.makedirs(os.path.join(os.path.dirname(out), "summ"), exist_ok=True)
open(os.path.join(os.path.dirname(out), "summ", "r03_smoother_mag_N8192_m512_sym_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
