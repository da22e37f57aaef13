#!/bin/bash
# usage (gpurun): tools/variants/cull_run.sh -- sweep / pixel-kernel time of config 3 and config 5 for the product library and every
# tools/variants/lib_*.so, twice, alternating; the product also with CELESTE_PIXEL_ORDER=colmajor (no ring tables)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for f in product product_colmajor product_rings tools/variants/lib_*.so; do
  L=""; ORD=""
  case $f in product) ;; product_colmajor) ORD=colmajor ;; product_rings) ORD=rings ;; *) L=$PWD/$f ;; esac
  CELESTE_PIXEL_ORDER=$ORD CELESTE_MI355X_LIB=$L python bench.py --no-config5 --no-cpu-baseline --no-live-pmc --no-extras --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c3 %-34s sweep %.4f prep %.4f pixel %.4f lift %.4f' % ('$f', d['ms_per_step'], d['kernels_ms']['prep'], d['kernels_ms']['pixel'], d['kernels_ms']['lift']))"
  CELESTE_PIXEL_ORDER=$ORD CELESTE_MI355X_LIB=$L python bench.py --config 5 --dtype f32 --steps 10 --warmup 2 --kernels-in-pass --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c5 %-34s value %.0f ms %.3f pixel f32 %.3f fp64 %.3f speedup %.3f err %s' % ('$f', d['value'], d['ms_per_step'], d['kernels_ms']['pixel'], d.get('fp64_pixel_kernel_ms',0), d.get('fp32_speedup_over_fp64_pixel_kernel',0), [round(d['fp32_vs_fp64_device'][k]*1e6,2) for k in 'vdh']))"
done; done
