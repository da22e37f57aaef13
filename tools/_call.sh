cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c21
B3="python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras"
B="python bench.py --config 5 --dtype f32 --steps 10 --warmup 2 --kernels-in-pass --no-cpu-baseline --no-extras"
for rep in 1 2; do
for v in new head; do
  if [ $v = new ]; then unset CELESTE_MI355X_LIB; else export CELESTE_MI355X_LIB=$GRAFT_REPO_ROOT/tools/variants/lib_$v.so; fi
  timeout 300 $B3 > gpurun_out/c21/f64_$v.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/c21/f64_$v.json'));print('f64 $v', round(d['value']), d['ms_per_step'], d['kernels_ms'])"
  timeout 300 $B > gpurun_out/c21/c5_$v.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/c21/c5_$v.json'));print('c5 $v', round(d['value']), d['ms_per_step'], d['kernels_ms'])"
done
done
unset CELESTE_MI355X_LIB
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 300 python bench.py --height 300 --width 260 --sources 60 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('layer us', d['optimizer']['cyclades_layer']['us_per_newton_iteration_of_the_slowest_target'])"
