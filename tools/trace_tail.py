#!/usr/bin/env python3
"""The step boundary from a rocprofv3 --kernel-trace csv: the last N kernels of the shortest step and the first N of the next one, every
stream, with start / end relative to the step's last Adam launch — what the main stream waits for between backward and forward."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows.sort(key=lambda r: int(r['Start_Timestamp']))
S = lambda r: int(r['Start_Timestamp']); E = lambda r: int(r['End_Timestamp'])
name = lambda r: re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('unsigned short', 'bf16')[:70]
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
# the LAST adam launch of each step (the optimizer update is split in two launches)
idx = [i for k, i in enumerate(idx) if k + 1 == len(idx) or S(rows[idx[k + 1]]) - E(rows[i]) > 2_000_000]
k = idx[len(idx) // 2]
t0 = S(rows[k])
streams = {}
for r in rows[max(0, k - 400):k + 400]:
    streams.setdefault(r['Stream_Id'], 0)
    streams[r['Stream_Id']] += E(r) - S(r)
order = sorted(streams, key=lambda s: -streams[s])
lab = {s: "main side s2 x x x".split()[i] if i < 3 else "x" for i, s in enumerate(order)}
print("time relative to the start of the step's last Adam launch; streams by busy time: " + ", ".join(f"{lab[s]}={s}" for s in order[:3]))
for r in rows[max(0, k - N):k + N]:
    print(f"{(S(r) - t0) / 1e3:9.1f} .. {(E(r) - t0) / 1e3:9.1f} us  {lab.get(r['Stream_Id'], '?'):5s} {name(r)}")
