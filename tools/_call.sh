cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c6
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/c6/pytest_gpu.txt
cat gpurun_out/c6/pytest_gpu.txt
timeout 600 python bench.py --no-config5 > gpurun_out/c6/bench.json 2> gpurun_out/c6/bench.err; tail -2 gpurun_out/c6/bench.err
python -c "
import json;d=json.load(open('gpurun_out/c6/bench.json'))
print(d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['roofline']['bound'])
for k in ('optimizer','joint_infer','single_call_latency_us','shard_projection','host_api_sources_per_sec'): print(k, d.get(k))
"
