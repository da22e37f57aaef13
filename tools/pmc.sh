#!/bin/bash
# usage: tools/pmc.sh <tag>   (run through gpurun); collects SQ counters for the bench field in separate passes
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/p1 -o p1 -- python $GRAFT_REPO_ROOT/tools/gpu_time.py > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p2 -o p2 -- python $GRAFT_REPO_ROOT/tools/gpu_time.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_SMEM --kernel-trace --output-format csv -d $OUT/p3 -o p3 -- python $GRAFT_REPO_ROOT/tools/gpu_time.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS --kernel-trace --output-format csv -d $OUT/p4 -o p4 -- python $GRAFT_REPO_ROOT/tools/gpu_time.py > /dev/null 2>&1
python - <<PY
import pandas as pd, glob
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    df = pd.read_csv(f)
    for k in ("pixel_kernel<2", "value_kernel", "lift_kernel"):
        d = df[df.Kernel_Name.str.contains(k, regex=False)]
        if len(d):
            print(k, d.groupby("Counter_Name").Counter_Value.mean().round(0).to_dict())
PY
