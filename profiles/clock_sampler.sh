#!/bin/bash
# Samples the shader clock, memory clock and package power every ~100 ms while the headline bench runs
# (the mid pass is half of every step): is the part clock- or power-limited during the timed region?
cd "$GRAFT_REPO_ROOT"
( python bench.py --steps 150 --warmup 2 --cpu-pairs 0 --skip-secondary > gpurun_out/clock_bench.json 2>/dev/null ) &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/GPU\[0\]\s*: //; s/clock level: //; s/Current Socket Graphics Package //' | tr '\n' ' '; echo
done > gpurun_out/clock_samples.txt
wait $BP
python -c "
import json; d=json.loads(open('gpurun_out/clock_bench.json').read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], {k:round(v['us_per_pair'],2) for k,v in d['kernels'].items()})"
awk 'NR%3==0' gpurun_out/clock_samples.txt | tail -40
