cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
timeout 900 python -m pytest tests/test_gpu_group.py tests/test_cabi_caller.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/c1/pytest.txt
cat gpurun_out/c1/pytest.txt
timeout 300 python bench.py --driver group --no-cpu-baseline --steps 200 > gpurun_out/c1/group1.json 2> gpurun_out/c1/group1.err; tail -3 gpurun_out/c1/group1.err; cat gpurun_out/c1/group1.json
timeout 300 python bench.py --driver group --gpus 2 --group-devices 0,0 --no-cpu-baseline --steps 100 > gpurun_out/c1/group2.json 2> gpurun_out/c1/group2.err; tail -3 gpurun_out/c1/group2.err; cat gpurun_out/c1/group2.json
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 > gpurun_out/c1/ranks1.json 2> gpurun_out/c1/ranks1.err; cat gpurun_out/c1/ranks1.json
timeout 900 tools/pmc_config5.sh c1_c5 2>&1 | tail -5
