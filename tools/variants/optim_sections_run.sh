#!/bin/bash
# usage (gpurun): tools/variants/optim_sections_run.sh -- optimiser per-iteration timing: product library, then every
# tools/variants/lib_optclk*.so (-DOPTIM_TIMING builds: per-section shader clocks of the step kernel)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/opt1
python tools/gpu_optim_sections.py 2>/dev/null | sed "s/^/product: /"
for f in tools/variants/clk/lib_optclk*.so; do
  CELESTE_MI355X_LIB=$PWD/$f python tools/gpu_optim_sections.py 2>/dev/null | sed "s/^/$(basename $f): /"
done
