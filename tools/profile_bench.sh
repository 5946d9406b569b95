#!/bin/bash
# Run on the GPU box (via gpurun): `rocprofv3 --kernel-trace --stats` of the headline bench command (the filter leg of
# BASELINE.json configs[2]; the extra legs are switched off so that the kernel table is the headline's) plus the default bench
# line itself.  Usage: tools/profile_bench.sh <tag> [steps]
set -u
TAG=${1:-r2}
STEPS=${2:-200}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT $REPO/gpurun_out/summ
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $REPO/bench.py --steps $STEPS --no-smoother --no-large --no-cpu-baseline > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
cd $REPO
python3 - "$OUT" "$TAG" "$STEPS" <<'PY'
import csv, glob, json, os, shutil, sys
out, tag, steps = sys.argv[1:4]
f = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
lines = [f"== rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {steps} --no-smoother --no-large --no-cpu-baseline =="]
if f:
    shutil.copy(f[0], os.path.join(os.path.dirname(out), "summ", f"{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    tot_ns, tot_calls = 0.0, 0
    for r in rows[:12]:
        lines.append(f"{r['Name'][:110]:110s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:10.1f} total_ms={float(r['TotalDurationNs'])/1e6:10.1f} pct={r['Percentage']}")
    sk = [r for r in rows if "step_kernel" in r["Name"]]
    tot_ns = sum(float(r["TotalDurationNs"]) for r in sk)
    lines.append(f"step_kernel: all variants together {tot_ns/1e6:.1f} ms over the run (timed + warm-up steps; an in-place flush is two dispatches per step)")
try:
    j = json.loads([l for l in open(os.path.join(out, "bench_profiled.json")) if l.startswith("{")][-1])
    r = j["roofline"]
    lines.append(f"bench line of the same run: value={j['value']:.0f} {j['unit']}, ms_per_step={j['ms_per_step']:.3f}, roofline.avg_launch_ms={r['avg_launch_ms']:.3f} (HIP events), "
                 f"scheduled_bytes_per_launch={r['scheduled_bytes_per_launch']:.4g}, achieved={r['achieved']:.0f} GB/s, frac={r['frac']:.3f}")
except Exception as exc:
    lines.append(f"bench line not parsed: {exc}")
open(os.path.join(os.path.dirname(out), "summ", f"{tag}_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
