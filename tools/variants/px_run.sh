#!/bin/bash
# usage (gpurun): tools/variants/px_run.sh -- pixel-kernel time of the bench sweep for the product library and every
# tools/variants/lib_px_*.so (twice each, alternating, to see the run-to-run noise)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for f in product tools/variants/lib_px_*.so; do
  L=""; [ $f != product ] && L=$PWD/$f
  CELESTE_MI355X_LIB=$L python bench.py --no-extras --no-cpu-baseline --no-live-pmc --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s pixel %.4f ms  sweep %.4f ms  prep %.4f lift %.4f' % ('$f', d['kernels_ms']['pixel'], d['ms_per_step'], d['kernels_ms']['prep'], d['kernels_ms']['lift']))"
done; done
