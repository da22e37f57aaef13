#!/bin/bash
# Read-only probe: does this lease expose more than one logical device (DPX/CPX compute partitions)?  Changes nothing.
out=gpurun_out/partition_probe.txt
{
echo "== rocm-smi --showcomputepartition"; rocm-smi --showcomputepartition 2>&1 | head -30
echo "== rocm-smi --showmemorypartition"; rocm-smi --showmemorypartition 2>&1 | head -30
echo "== amd-smi partition"; (amd-smi partition 2>&1 || true) | head -60
echo "== amd-smi list"; (amd-smi list 2>&1 || true) | head -40
echo "== rocminfo agents"; rocminfo 2>&1 | grep -E "Marketing Name|Compute Unit|Name: +gfx|Uuid" | head -40
echo "== hipGetDeviceCount"; python - <<'PY'
import torch
print("device_count", torch.cuda.device_count())
for i in range(torch.cuda.device_count()):
    p = torch.cuda.get_device_properties(i)
    print(i, p.name, p.multi_processor_count, p.total_memory)
PY
echo "== sysfs partition files"; for f in /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/available_compute_partition /sys/class/drm/card*/device/current_memory_partition; do [ -e "$f" ] && { echo "$f: $(cat $f 2>&1)"; ls -l $f; }; done
echo "== id"; id
echo "== env"; env | grep -E "HIP_VISIBLE|ROCR_VISIBLE|CUDA_VISIBLE|HSA_" 
echo "== sclk"; rocm-smi --showclocks 2>&1 | head -30
} > $out 2>&1
cat $out
