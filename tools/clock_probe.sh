#!/bin/bash
# Shader / memory clocks and power while the complete T = 3000 filter run of the headline configuration is in flight (sampled every 2 s).
# Run on the GPU box:  bash tools/clock_probe.sh > gpurun_out/clock_probe.txt
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-smoother --no-large --no-traffic > /tmp/clk_bench.json 2> /tmp/clk_bench.err &
BP=$!
for i in $(seq 1 30); do
  sleep 2
  if ! kill -0 $BP 2>/dev/null; then break; fi
  echo "t=$((2*i))s $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|Power' | tr -s ' ' | tr '\n' ';')"
done
wait $BP
python3 -c "
import json
j=json.loads([l for l in open('/tmp/clk_bench.json') if l.startswith('{')][-1])
print('headline', j['value']/1e6, 'M; complete run', j['filter_full_T']['seconds'], 's', [round(w['kernel_ms_per_step'],2) for w in j['filter_full_T']['windows']])
"
