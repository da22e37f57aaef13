#!/bin/bash
# usage: tools/pmc2.sh <tag> "<counters...>"  -> prints mean per-kernel counter values (bench field)
TAG=${1:-pmcx}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $@ --kernel-trace --output-format csv -d $OUT -o p -- python $GRAFT_REPO_ROOT/tools/gpu_time.py > /dev/null 2>&1
python - <<PY
import pandas as pd, glob
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    df = pd.read_csv(f)
    for k in ("pixel_kernel", "value_kernel", "lift_kernel"):
        d = df[df.Kernel_Name.str.contains(k)]
        if len(d):
            print(k, d.groupby("Counter_Name").Counter_Value.mean().round(0).to_dict())
PY
