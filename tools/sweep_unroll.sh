#!/bin/bash
# Tuning aid (GPU box): filter throughput of a few sizes for the current build / RBPF_LIB_PATH variant.
run() { python bench.py $2 --no-cpu-baseline --no-smoother --no-large > gpurun_out/sw.log 2>&1; python -c "
import json
l=[x for x in open('gpurun_out/sw.log') if x.startswith('{')]
print('$1', round(json.loads(l[-1])['value']) if l else 'FAILED')"; }
run "fp32 m256 N8192" "--storage fp32 --steps 600"
run "fp32 m512 N65536" "--storage fp32 --particles 65536 --m 512 --steps 45 --warmup 6"
run "fp32 m1024 N32768 lazy2" "--storage fp32 --particles 32768 --m 1024 --lazy-depth 2 --steps 30 --warmup 4"
run "fp64 m256 N8192" "--steps 600"
run "fp64 m512 N65536" "--particles 65536 --m 512 --steps 45 --warmup 6"
