#!/usr/bin/env python3
"""Kernels of one step around a named kernel (rocprofv3 --kernel-trace csv): usage trace_region.py trace.csv kernel-substring before after"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat, nb, na = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rows.sort(key=lambda r: int(r['Start_Timestamp']))
S = lambda r: int(r['Start_Timestamp']); E = lambda r: int(r['End_Timestamp'])
name = lambda r: re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '')).replace('void ', '').replace('unsigned short', 'bf16')[:70]
hits = [i for i, r in enumerate(rows) if pat in r['Kernel_Name']]
k = hits[len(hits) // 2]
t0 = S(rows[k])
busy = {}
for r in rows[max(0, k - 600):k + 600]:
    busy[r['Stream_Id']] = busy.get(r['Stream_Id'], 0) + E(r) - S(r)
order = sorted(busy, key=lambda s: -busy[s])
lab = {s: ("main", "side", "s2")[i] if i < 3 else "x" for i, s in enumerate(order)}
prev_end = {}
for r in rows[max(0, k - nb):k + na]:
    s = r['Stream_Id']
    gap = (S(r) - prev_end[s]) / 1e3 if s in prev_end else 0.0
    prev_end[s] = E(r)
    print(f"{(S(r) - t0) / 1e3:9.1f} .. {(E(r) - t0) / 1e3:9.1f} us  ({(E(r) - S(r)) / 1e3:6.1f})  gap {gap:6.1f}  {lab.get(s, '?'):5s} {name(r)}")
