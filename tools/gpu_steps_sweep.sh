#!/bin/bash
# usage (gpurun): tools/gpu_steps_sweep.sh -- the headline line for several --steps (fresh process each): how much of a short run is the chip's clock ramp
cd $GRAFT_REPO_ROOT
for k in 20 50 100 200 500 1000; do
  python bench.py --steps $k --warmup 5 --no-extras --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['step_ms']
print('--steps %4d --warmup 5: %.3f M sources/s, ms_per_step %.4f | step_ms first %.3f p50 %.3f last %.3f | sclk after the loop %s MHz' % ($k, d['value']/1e6, d['ms_per_step'], s['first'], s['p50'], s['last'], d['sclk_mhz_after_loop']))"
done
