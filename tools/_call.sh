cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_group.py -m gpu -x -q -k "bench_group" 2>&1 | grep -vE "RCCL|HIP version|ROCm|Hostname|Librccl" | tail -15
