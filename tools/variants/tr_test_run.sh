#!/bin/bash
# usage (gpurun): the sub-problem accuracy table (tests/test_gpu_tr_subproblem.py -s) with the product library and every variant
cd $GRAFT_REPO_ROOT
one() { timeout 600 python -m pytest tests/test_gpu_tr_subproblem.py -q -m gpu -s -k "digits or hessians" 2>&1 | grep -E "error / bound|passed|failed" | cut -c1-170 | sed "s|^|$1: |"; }
one product
for f in tools/variants/lib_*.so; do CELESTE_MI355X_LIB=$PWD/$f one $(basename $f); done
