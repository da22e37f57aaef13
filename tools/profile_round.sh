#!/bin/bash
# usage (through gpurun): tools/profile_round.sh <tag>
# Writes gpurun_out/<tag>/: bench.json, kernel-trace stats (csv) of the same bench command, and the
# HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in separate passes, guide section "HBM") plus one SQ pass.
# Summarise locally afterwards with: python tools/summarize_profile.py <tag>
TAG=${1:-r01}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-live-pmc --no-config5 --no-variable-psf > $OUT/bench_traced.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-live-pmc --no-config5 --no-variable-psf > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-live-pmc --no-config5 --no-variable-psf > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o sq -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-live-pmc --no-config5 --no-variable-psf > /dev/null 2> $OUT/pmc_sq.err
# the kernels that dominate a run with extras -- the optimiser's lock-step step kernel, the persistent fused launch of a
# Cyclades layer and the joint-inference dataflow launch -- alone under the counters (bench.py --pmc-child optim)
OPT="python $ROOT/bench.py --pmc-child optim --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-live-pmc --no-config5 --no-variable-psf"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/opt_trace -o trace -- $OPT > /dev/null 2> $OUT/opt_trace.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/opt_sq -o sq -- $OPT > /dev/null 2> $OUT/opt_sq.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/opt_lds -o lds -- $OPT > /dev/null 2> $OUT/opt_lds.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/opt_fetch -o fetch -- $OPT > /dev/null 2> $OUT/opt_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/opt_write -o write -- $OPT > /dev/null 2> $OUT/opt_write.err
# keep the merged output small: drop the per-dispatch traces of the PMC passes' kernel-trace
find $OUT -name "*kernel_trace.csv" \( -path "*pmc_*" -o -path "*opt_sq*" -o -path "*opt_lds*" -o -path "*opt_fetch*" -o -path "*opt_write*" \) -delete
ls -la $OUT $OUT/trace | head -30
cat $OUT/bench.json
