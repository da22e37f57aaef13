cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5
B="python bench.py --config 5 --dtype f32 --steps 10 --warmup 2 --kernels-in-pass --no-cpu-baseline --no-extras"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/c5/$name.json 2> gpurun_out/c5/$name.err; python -c "import json;d=json.load(open('gpurun_out/c5/$name.json'));print('$name', round(d['value']), d['kernels_ms'], d['fp32_vs_fp64_device'])" || tail -3 gpurun_out/c5/$name.err; }
run new X=1
run prev CELESTE_MI355X_LIB=$GRAFT_REPO_ROOT/tools/variants/lib_prev.so
run new_again X=1
B3="python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-extras"
for e in CELESTE_CHUNK_GROUP=1 CELESTE_CHUNK_GROUP=2 CELESTE_CHUNK_GROUP=3 CELESTE_CHUNK_GROUP=4 CELESTE_CHUNK_GROUP=2; do
  env $e timeout 300 $B3 > gpurun_out/c5/f64_$e.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/c5/f64_$e.json'));print('f64 $e', round(d['value']), d['kernels_ms'])"
done
