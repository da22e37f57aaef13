#!/bin/bash
# usage (gpurun): the joint-inference parity test's numbers with the product library and every tools/variants/lib_*.so
cd $GRAFT_REPO_ROOT
one() { timeout 300 python -m pytest tests/test_gpu_round2.py -q -m gpu -s -k "joint_inference_on_the_device or randomised" 2>&1 | grep -E "joint inference,|largest difference|passed|failed|max abs" | sed "s/^/$1: /"; }
one product
for f in tools/variants/lib_*.so; do CELESTE_MI355X_LIB=$PWD/$f one $(basename $f); done
