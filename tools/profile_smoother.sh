#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats of the information-form smoother at the per-GPU share of the metric
# configuration (N_P = 8192, m = 512), or of any other smoother_bench.py run.
# Usage: tools/profile_smoother.sh <tag> <T> [key=value ...]            dense-mag N_P = 8192, m = 512, N_K = 2, T steps
#        tools/profile_smoother.sh <tag> -- <smoother_bench.py args>    anything else
set -u
TAG=${1:-r2}
if [ "${2:-}" = "--" ]; then shift 2; ARGS="$*"; else T=${2:-120}; shift 2; ARGS="mag 8192 $T 512 2 info $*"; fi
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $REPO/tools/smoother_bench.py $ARGS > $OUT/run.log 2>&1
cd $REPO
python3 - "$OUT" "$TAG" "$ARGS" <<'PY'
import csv, glob, os, sys
out, tag, opts = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
lines = [f"== rocprofv3 --kernel-trace --stats: tools/smoother_bench.py {opts} =="]
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    for r in rows[:14]:
        lines.append(f"{r['Name'][:110]:110s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:10.1f} total_ms={float(r['TotalDurationNs'])/1e6:10.1f} pct={r['Percentage']}")
lines += [l.strip() for l in open(os.path.join(out, "run.log")) if l.startswith("{")]
os.makedirs(os.path.join(os.path.dirname(out), "summ"), exist_ok=True)
open(os.path.join(os.path.dirname(out), "summ", f"{tag}_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
