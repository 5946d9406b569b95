#!/bin/bash
# r05 profiles, run on the GPU box via gpurun: rocprofv3 kernel stats of (1) the DEFAULT headline path (the filter leg of bench.py), (1b) configs[4]'s share on fp32 block-lower tiles,
# (2) the information-form smoother at the per-GPU share with the library defaults (carried factors) and with chol_refresh = 1, and
# (3) HBM counters of the default smoother (separate --pmc passes; no trace domains with --pmc).  Summaries -> gpurun_out/summ/.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r05_prof
mkdir -p $OUT $REPO/gpurun_out/summ
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/filter -o t -- python3 $REPO/bench.py --steps 200 --no-smoother --no-large --no-cpu-baseline --no-filter-full > $OUT/filter.json 2> $OUT/filter.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -o t -- python3 $REPO/tools/sym16_trace.py fp32sym 4 32768 60 > $OUT/c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sm_default -o sm -- python3 $REPO/tools/smoother_bench.py mag 8192 130 512 2 info lazy_depth=3 storage=2 > $OUT/sm_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sm_fresh -o sm -- python3 $REPO/tools/smoother_bench.py mag 8192 40 512 2 info lazy_depth=3 storage=2 chol_refresh=1 > $OUT/sm_fresh.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $REPO/tools/smoother_bench.py mag 8192 40 512 2 info lazy_depth=3 storage=2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $REPO/tools/smoother_bench.py mag 8192 40 512 2 info lazy_depth=3 storage=2 > $OUT/pmc_write.log 2>&1
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
summ = os.path.join(os.path.dirname(out), "summ")
def stats(leg, title, n=14):
    lines = [f"== rocprofv3 --kernel-trace --stats: {title} =="]
    f = glob.glob(os.path.join(out, leg, "**", "*kernel_stats.csv"), recursive=True)
    if not f:
        return lines + ["no stats file"]
    for r in list(csv.DictReader(open(f[0])))[:n]:
        lines.append(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s}  avg {float(r['AverageNs'])/1e6:9.3f} ms  total {float(r['TotalDurationNs'])/1e6:10.2f} ms  {float(r['Percentage']):6.2f} %")
    return lines
lines = stats("filter", "python3 bench.py --steps 200 --no-smoother --no-large --no-cpu-baseline --no-filter-full (the DEFAULT headline path, r05)")
try:
    j = json.loads([l for l in open(os.path.join(out, "filter.json")) if l.startswith("{")][-1])
    r = j["roofline"]
    lines.append(f"bench line of the same run: value={j['value']:.0f} {j['unit']}, ms_per_step={j['ms_per_step']:.3f}, roofline.avg_launch_ms={r['avg_launch_ms']:.3f} (HIP events), "
                 f"scheduled_bytes_per_launch={r['scheduled_bytes_per_launch']:.4g}, achieved={r['achieved']:.0f} GB/s, frac={r['frac']:.3f}")
except Exception as exc:
    lines.append(f"bench line not parsed: {exc}")
open(os.path.join(summ, "r05_filter_sym_N65536_m512_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
lines = stats("c4", "tools/sym16_trace.py fp32sym 4 32768 60 (BASELINE.json configs[4]'s per-GPU share: N = 32768, m = 1024, fp32 block-lower tiles, lazy_depth 4)", 8)
try:
    lines.append("the run's own line: " + [l for l in open(os.path.join(out, "c4.log")) if "particle-steps/s" in l][-1].strip())
except Exception as exc:
    lines.append(f"run line not found: {exc}")
open(os.path.join(summ, "r05_filter_fp32sym_N32768_m1024_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
lines = stats("sm_default", "tools/smoother_bench.py mag 8192 130 512 2 info lazy_depth=3 storage=2 (library defaults: carried factors, K = 32)")
lines += stats("sm_fresh", "tools/smoother_bench.py mag 8192 40 512 2 info lazy_depth=3 storage=2 chol_refresh=1 (the reference's arithmetic)", 8)
lines.append("== rocprofv3 --pmc (separate passes), means per dispatch, default smoother T = 40 ==")
agg = defaultdict(lambda: [0.0, 0])
for leg in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(os.path.join(out, leg, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            name = "chol_sweep_kernel" if "chol_sweep_kernel" in k else "step_sym_kernel (all variants)" if "step_sym_kernel" in k else None
            if name:
                a = agg[(name, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (name, ctr), (s, c) in sorted(agg.items()):
    lines.append(f"{name:34s} {ctr:12s} mean {s / c * 1024 / 1e9:8.3f} GB per dispatch  ({c} dispatches; KiB counter; FETCH_SIZE counts half of wide reads on gfx950)")
if ("chol_sweep_kernel", "FETCH_SIZE") in agg and ("chol_sweep_kernel", "WRITE_SIZE") in agg:
    fe = agg[("chol_sweep_kernel", "FETCH_SIZE")]; wr = agg[("chol_sweep_kernel", "WRITE_SIZE")]
    tr = 2 * fe[0] / fe[1] * 1024 + wr[0] / wr[1] * 1024
    lines.append(f"chol_sweep_kernel: 2 x FETCH + WRITE = {tr / 1e9:.2f} GB per launch of 8192 particles (algorithmic 2 x 1.213 MB x 8192 = 19.87 GB)")
open(os.path.join(summ, "r05_smoother_mag_N8192_m512_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
