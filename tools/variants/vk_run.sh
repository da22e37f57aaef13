#!/bin/bash
# usage (gpurun): tools/variants/vk_run.sh -- value_kernel occupancy variants: prep bucket (list + tables + neighbours' light) and step, config 3 and config 5
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for f in product tools/variants/lib_value_w*.so; do
  L=""; [ $f != product ] && L=$PWD/$f
  CELESTE_MI355X_LIB=$L python bench.py --no-config5 --no-cpu-baseline --no-live-pmc --no-variable-psf --no-extras --steps 300 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s config3 step %.4f prep %.4f pixel %.4f lift %.4f' % ('$f', d['ms_per_step'], d['kernels_ms']['prep'], d['kernels_ms']['pixel'], d['kernels_ms']['lift']))"
  CELESTE_MI355X_LIB=$L python bench.py --config 5 --dtype f32 --steps 10 --warmup 2 --kernels-in-pass --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s config5 step %.3f prep %.3f pixel %.3f lift %.3f' % ('$f', d['ms_per_step'], d['kernels_ms']['prep'], d['kernels_ms']['pixel'], d['kernels_ms']['lift']))"
done; done
