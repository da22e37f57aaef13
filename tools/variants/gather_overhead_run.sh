#!/bin/bash
# usage (gpurun): host overhead of the catalog-gather path on a small sweep: one rank under the launcher, without and with the
# RCCL gather (CELESTE_GATHER_SINGLE=1); also shows that rank 0's stdout is the JSON line alone
cd $GRAFT_REPO_ROOT
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --sources 250 --steps 300 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null > /tmp/out.txt; echo "$1: stdout lines $(wc -l < /tmp/out.txt)"; python -c "import sys,json; d=json.loads(open('/tmp/out.txt').read()); print('   ms_per_step', d['ms_per_step'], d['kernels_ms'], d['config']['gather_backend'])"; }
run "no gather "
CELESTE_GATHER_SINGLE=1 run "rccl gather"
