#!/bin/bash
# usage (gpurun): tools/variants/run_optim.sh [reps] -- optimiser timing with the product library and every variant
cd $GRAFT_REPO_ROOT
one() { python tools/gpu_optim_time.py 2>/dev/null | grep "max_iters 50" | sed "s/^/$1: /"; }
for rep in $(seq 1 ${1:-2}); do
one product
for f in tools/variants/lib_*.so; do CELESTE_MI355X_LIB=$PWD/$f one $(basename $f); done
done
