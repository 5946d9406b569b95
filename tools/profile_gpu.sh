#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for the bench workload.
# Usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 200 --warmup 20 --no-cpu-baseline --no-smoother --no-large}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/bench_pmc_write.log 2>&1
cd $REPO
find $OUT -name "*.csv" | head -20
python3 tools/summarise_profile.py $OUT $TAG
