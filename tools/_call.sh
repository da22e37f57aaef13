cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_group.py tests/test_cabi_caller.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2; done
