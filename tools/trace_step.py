#!/usr/bin/env python3
"""Per-stream composition of ONE training step from a rocprofv3 --kernel-trace csv (step = between two adam launches)."""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
# the optimizer update is two launches (encoder half, then the overlapped upper half): keep the LAST adam launch of each step
idx = [i for k, i in enumerate(idx) if k + 1 == len(idx) or int(rows[idx[k + 1]]['Start_Timestamp']) - int(rows[i]['End_Timestamp']) > 2_000_000]
# the shortest step of the run (bench.py's instrumented pass, which synchronises per launch, comes last and is skipped)
spans = [(int(rows[idx[i + 1]]['End_Timestamp']) - int(rows[idx[i] + 1]['Start_Timestamp']), i) for i in range(len(idx) - 1)]
_, best = min(spans)
a, b = idx[best], idx[best + 1]
step = rows[a + 1:b + 1]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
busy, (cs, ce) = 0, iv[0]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"step span {(t1 - t0) / 1e6:.3f} ms, busy union {busy / 1e6:.3f} ms, sum of kernels {sum(dur(r) for r in step) / 1e3:.3f} ms, launches {len(step)}")
streams = sorted({r['Stream_Id'] for r in step})
for sid in streams:
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in step:
        if r['Stream_Id'] != sid:
            continue
        n = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '')).replace('void ', '').replace('unsigned short', 'bf16')
        d[n][0] += 1; d[n][1] += dur(r)
    print(f"--- stream {sid}: {sum(v[1] for v in d.values()) / 1e3:.3f} ms")
    for k, v in sorted(d.items(), key=lambda x: -x[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
        print(f"  {k:62s} {v[0]:4d} {v[1]:8.1f} us")
