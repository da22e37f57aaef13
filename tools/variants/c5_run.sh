#!/bin/bash
# usage (gpurun): tools/variants/c5_run.sh -- config 5 (fp32 mode) for the product library and every tools/variants/lib_*.so, twice, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for f in product tools/variants/lib_*.so; do
  L=""; [ $f != product ] && L=$PWD/$f
  CELESTE_MI355X_LIB=$L python bench.py --config 5 --dtype f32 --steps 10 --warmup 2 --kernels-in-pass --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-30s value %.0f ms %.3f pixel f32 %.3f fp64 %.3f speedup %.3f err %s' % ('$f', d['value'], d['ms_per_step'], d['kernels_ms']['pixel'], d.get('fp64_pixel_kernel_ms',0), d.get('fp32_speedup_over_fp64_pixel_kernel',0), [round(d['fp32_vs_fp64_device'][k]*1e6,2) for k in 'vdh']))"
done; done
