#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the information-form smoother at the per-GPU share of the
# N=65536 configuration (N_P=8192, m=512) over a short horizon.
# Usage: tools/profile_smoother.sh <tag> [smoother_bench args...]
set -u
TAG=${1:-r01s}; shift || true
ARGS=${@:-mag 8192 40 512 2 info}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/tools/smoother_bench.py $ARGS > $OUT/smoother_trace.log 2>&1
cd $REPO
python3 tools/summarise_profile.py $OUT $TAG
