#!/bin/bash
# usage (gpurun): tools/shard_trace.sh -- kernel trace of tools/gpu_shard_time.py -> gpurun_out/shard/s_kernel_trace.csv
cd /tmp && export TMPDIR=/tmp; rm -rf $GRAFT_REPO_ROOT/gpurun_out/shard; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/shard
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/shard -o s -- python $GRAFT_REPO_ROOT/tools/gpu_shard_time.py > /dev/null 2>&1
