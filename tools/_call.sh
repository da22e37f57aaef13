cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c19
B="python bench.py --config 5 --dtype f32 --steps 10 --warmup 2 --kernels-in-pass --no-cpu-baseline --no-extras"
for rep in 1 2; do
for e in X=1 CELESTE_FP32_CHUNK_256=1; do
  env $e timeout 300 $B > gpurun_out/c19/c5_$e.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/c19/c5_$e.json'));print('c5 $e', round(d['value']), d['ms_per_step'], d['kernels_ms'], d['fp32_vs_fp64_device'])"
done
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_round2.py tests/test_gpu_group.py tests/test_mutants.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
