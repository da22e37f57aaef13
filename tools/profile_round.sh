#!/bin/bash
# usage (through gpurun): tools/profile_round.sh <tag>
# Writes gpurun_out/<tag>/: bench.json, kernel-trace stats (csv) of the same bench command, and the
# HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in separate passes, guide section "HBM").
TAG=${1:-r01}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_traced.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
python - <<PY
import pandas as pd, glob, json
for f in glob.glob("$OUT/trace/*kernel_stats.csv"):
    print(open(f).read()[:3000])
for tag in ("fetch", "write"):
    for f in glob.glob("$OUT/pmc_%s/*counter_collection.csv" % tag):
        df = pd.read_csv(f)
        print(tag, df.groupby(["Kernel_Name", "Counter_Name"]).Counter_Value.mean().to_string()[:2000])
PY
cat $OUT/bench.json
