"""Summarise gpurun_out/shard/s_kernel_trace.csv (tools/shard_trace.sh): average duration per kernel and grid, and the
timeline of one steady-state sweep per shard size."""
import csv, collections, sys
rows = list(csv.DictReader(open('/root/repo/gpurun_out/shard/s_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r['Kernel_Name'].split('(')[0][:36], int(r['Grid_Size_X']), int(r['Workgroup_Size_X']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    if len(v) >= 20: print("%-38s grid %8d wg %4d calls %4d avg %7.1f us" % (k[0], k[1], k[2], len(v), sum(v) / len(v)))
for n in (2000, 1000, 500, 250, 125):
    idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('lift_kernel') and int(r['Grid_Size_X']) in (n * 512, n * 256)]
    if len(idx) < 40: continue
    i = idx[30]; j = idx[29]
    t0 = int(rows[j + 1]['Start_Timestamp'])
    print("-- %d targets" % n)
    for r in rows[j + 1:i + 1]:
        print("   %-30s start %7.1f dur %6.1f" % (r['Kernel_Name'].split('(')[0][:30], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
