#!/usr/bin/env python3
"""kernels of every stream around each launch of a named kernel inside ONE step of a rocprofv3 --kernel-trace csv
   usage: trace_window.py <kernel_trace.csv> <name substring> [n windows]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
idx = [i for k, i in enumerate(idx) if k + 1 == len(idx) or int(rows[idx[k + 1]]['Start_Timestamp']) - int(rows[i]['End_Timestamp']) > 2_000_000]
spans = [(int(rows[idx[i + 1]]['End_Timestamp']) - int(rows[idx[i] + 1]['Start_Timestamp']), i) for i in range(len(idx) - 1)]
_, best = min(spans)
step = rows[idx[best] + 1:idx[best + 1] + 1]
t0 = int(step[0]['Start_Timestamp'])
short = lambda n: re.sub(r'\(.*', '', n.replace('(anonymous namespace)::', '')).replace('void ', '').replace('unsigned short', 'bf16')[:58]
hits = [r for r in step if sys.argv[2] in r['Kernel_Name']]
for h in hits[:int(sys.argv[3]) if len(sys.argv) > 3 else 3]:
    s, e = int(h['Start_Timestamp']), int(h['End_Timestamp'])
    print(f"== {short(h['Kernel_Name'])} on stream {h['Stream_Id']}: {(s - t0) / 1e3:.1f} .. {(e - t0) / 1e3:.1f} us ({(e - s) / 1e3:.1f} us), grid {h.get('Grid_Size_X', '?')} wg {h.get('Workgroup_Size_X', '?')}")
    for r in step:
        rs, re_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if re_ > s - 30000 and rs < e + 30000:
            print(f"     stream {r['Stream_Id']} {(rs - t0) / 1e3:9.1f} .. {(re_ - t0) / 1e3:9.1f}  {(re_ - rs) / 1e3:7.1f} us  {short(r['Kernel_Name'])}  grid {r.get('Grid_Size_X', '?')}")
