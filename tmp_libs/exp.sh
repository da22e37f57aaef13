cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_optimizer.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], round(d['value']), round(d['grad_only_sources_per_sec_rank0']), d['split_variant']['kernel_ms'], round(d['optimizer']['optimized_sources_per_sec']))"; done
