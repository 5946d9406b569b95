#!/bin/bash
# kernel-trace averages of the step kernels for the product library and tuning libraries, same box:  bash tools/kt.sh STEPS [lib-suffix ...]
R=$(pwd); S=${1:-100}; shift
export TMPDIR=/tmp; cd /tmp
for L in product "$@"; do
  if [ $L != product ]; then export RBPF_LIB_PATH=$R/rao-blackwellized-slam-smoothing_amd/lib/librbpf_hip_$L.so; fi
  rm -rf /tmp/kt_$L
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$L -o t -- python3 $R/bench.py --steps $S --warmup 12 --no-cpu-baseline --no-smoother --no-large --no-traffic --no-filter-full > /tmp/kt_$L.json 2>/dev/null
  F=$(find /tmp/kt_$L -name "*kernel_stats.csv")
  python3 - "$F" $L <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "step_sym_kernel" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(sys.argv[2], " ".join(f"{r['Name'].split('<')[1].split('>')[0].replace('double, ', '')}:{float(r['AverageNs'])/1e6:.3f}" for r in rows), f"| total {tot/1e6:.1f} ms")
PY
done
