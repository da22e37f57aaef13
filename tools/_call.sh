cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c18
B="python bench.py --config 5 --dtype f32 --steps 10 --warmup 2 --kernels-in-pass --no-cpu-baseline --no-extras"
for rep in 1 2; do
for v in new head; do
  if [ $v = new ]; then unset CELESTE_MI355X_LIB; else export CELESTE_MI355X_LIB=$GRAFT_REPO_ROOT/tools/variants/lib_$v.so; fi
  timeout 300 $B > gpurun_out/c18/c5_$v.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/c18/c5_$v.json'));print('c5 $v', round(d['value']), d['ms_per_step'], d['kernels_ms'], d['fp32_vs_fp64_device'])"
done
done
unset CELESTE_MI355X_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_round2.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
