#!/bin/bash
# usage (gpurun): tools/variants/ab_run.sh -- the bench's secondary figures for the product library and every tools/variants/lib_*.so, twice, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for f in product tools/variants/lib_*.so; do
  L=""; [ $f != product ] && L=$PWD/$f
  CELESTE_MI355X_LIB=$L python bench.py --no-config5 --no-cpu-baseline --no-live-pmc --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
o=d['optimizer']
print('%-34s sweep %.4f pixel %.4f | layer %.1f us/it joint %.4f s layered %.3f opt %.0f/s | single call %.1f us | shard8 %.4f shard4 %.4f' % ('$f', d['ms_per_step'], d['kernels_ms']['pixel'], o['cyclades_layer']['us_per_newton_iteration_of_the_slowest_target'], d['joint_infer']['c_call_seconds'], d['joint_infer']['layer_by_layer_seconds'], o['optimized_sources_per_sec'], d['single_call_latency_us']['median'], d['shard_projection']['8']['ms_per_sweep'], d['shard_projection']['4']['ms_per_sweep']))"
done; done
