cd $GRAFT_REPO_ROOT
timeout 1500 bash tools/profile_round.sh r05d > gpurun_out/r05d.log 2>&1; tail -1 gpurun_out/r05d.log | cut -c1-300
timeout 600 bash tools/pmc_config5.sh r05d_c5 > gpurun_out/r05d_c5.log 2>&1
timeout 300 python bench.py --driver group --config 5 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r05d_group_c5.json 2> gpurun_out/r05d_group_c5.err; python -c "import json;d=json.load(open('gpurun_out/r05d_group_c5.json'));print('group c5', round(d['value']), d['ms_per_step'], d['kernels_ms'], d['config']['member_gather_ms'])"
timeout 300 python bench.py --driver group --steps 200 --no-cpu-baseline > gpurun_out/r05d_group_c3.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r05d_group_c3.json'));print('group c3', round(d['value']), d['ms_per_step'], d['kernels_ms'], d['config']['member_gather_ms'])"
