cd $GRAFT_REPO_ROOT
timeout 1500 bash tools/profile_round.sh r05b > gpurun_out/r05b.log 2>&1; tail -3 gpurun_out/r05b.log
timeout 600 bash tools/pmc_config5.sh r05b_c5 > gpurun_out/r05b_c5.log 2>&1; tail -2 gpurun_out/r05b_c5.log
