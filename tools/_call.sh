cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c7
B3="python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras"
for e in X=1 CELESTE_NO_SIDE_STREAM=1 X=2 CELESTE_NO_SIDE_STREAM=1; do
  env $e timeout 300 $B3 > gpurun_out/c7/f64_$e.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/c7/f64_$e.json'));print('f64 $e', round(d['value']), d['ms_per_step'], d['kernels_ms'])"
done
B="python bench.py --config 5 --dtype f32 --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
for e in X=1 CELESTE_NO_SIDE_STREAM=1; do
  env $e timeout 300 $B > gpurun_out/c7/c5_$e.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/c7/c5_$e.json'));print('c5 $e', round(d['value']), d['ms_per_step'], d['kernels_ms'])"
done
timeout 900 python -m pytest tests/test_mutants.py tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q 2>&1 | tail -5
