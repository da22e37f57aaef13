#!/bin/bash
# usage (through gpurun): tools/pmc_config5.sh <tag>
# Counter passes for the single-precision kernels of BASELINE configs[4] (pixel_kernel<2, float>, value_kernel<float, .>) at
# full size: 16 fields, 80 images, 30 000 sources.  Each counter set in its own pass with --kernel-trace only (guide, HBM
# section).  Summarise locally with: python tools/summarize_config5.py <tag>
TAG=${1:-r05c5}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --config 5 --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-live-pmc"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/bench_traced.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o sq -- $CMD > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc_lds -o lds -- $CMD > /dev/null 2> $OUT/pmc_lds.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- $CMD > /dev/null 2> $OUT/pmc_write.err
find $OUT -name "*kernel_trace.csv" -path "*pmc_*" -delete
find $OUT/trace -name "*kernel_trace.csv" -delete
ls -la $OUT | head
