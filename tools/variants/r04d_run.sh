#!/bin/bash
# round 4, call d: full GPU suite, the bench line, small-shard A/B (prep folded into the list launch, four-wave value items),
# config 5 with and without the fp32 row fold
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.log
python bench.py --no-config5 --no-cpu-baseline --no-live-pmc > $O/bench.json 2> $O/bench.err
shards() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
print('$2: value %.0f ms %.4f kernels %s' % (d['value'], d['ms_per_step'], d['kernels_ms']), {k:(round(v['ms_per_sweep'],4), round(v.get('compute_bound_efficiency',1),3)) for k,v in d['shard_projection'].items() if k!='note'})
print('   cyclades', d['optimizer']['cyclades_layer']['us_per_newton_iteration_of_the_slowest_target'], 'joint', d['joint_infer']['seconds'], 'opt', d['optimizer']['optimized_sources_per_sec'])
"; }
shards $O/bench.json product
CELESTE_NO_WIDE_VALUE=1 python bench.py --no-config5 --no-cpu-baseline --no-live-pmc > $O/bench_nowide.json 2> /dev/null; shards $O/bench_nowide.json no-wide-value
CELESTE_NO_FUSED_PREP=1 python bench.py --no-config5 --no-cpu-baseline --no-live-pmc > $O/bench_noprep.json 2> /dev/null; shards $O/bench_noprep.json no-fused-prep
for f in product tools/variants/lib_f32_norow.so; do
  L=""; [ $f != product ] && L=$PWD/$f
  CELESTE_MI355X_LIB=$L python bench.py --config 5 --dtype f32 --steps 10 --warmup 2 --kernels-in-pass --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$f config5 f32: value %.0f ms %.3f kernels %s fp64 pixel %.3f speedup %.3f err %s' % (d['value'], d['ms_per_step'], d['kernels_ms'], d.get('fp64_pixel_kernel_ms',0), d.get('fp32_speedup_over_fp64_pixel_kernel',0), d.get('fp32_vs_fp64_device')))"
done 2>&1 | tee $O/config5.txt
