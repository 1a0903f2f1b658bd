#!/usr/bin/env python3
"""Where does the step's wall time go?  From a rocprofv3 --kernel-trace csv: for the shortest step, the timeline is cut into
segments by every kernel start/end; each segment is attributed to the set of streams busy in it.  Prints the time with
0 / 1 / 2 / 3 streams busy, the idle gaps on the main stream (count, total, largest with their neighbours) and, per kernel
family on the main stream, how much of its run time had another stream's kernel running beside it."""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
idx = [i for k, i in enumerate(idx) if k + 1 == len(idx) or int(rows[idx[k + 1]]['Start_Timestamp']) - int(rows[i]['End_Timestamp']) > 2_000_000]
spans = [(int(rows[idx[i + 1]]['End_Timestamp']) - int(rows[idx[i] + 1]['Start_Timestamp']), i) for i in range(len(idx) - 1)]
_, best = min(spans)
step = rows[idx[best] + 1: idx[best + 1] + 1]
S = lambda r: int(r['Start_Timestamp']); E = lambda r: int(r['End_Timestamp'])
name = lambda r: re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '')).replace('void ', '').replace('unsigned short', 'bf16')[:60]
t0, t1 = S(step[0]), max(E(r) for r in step)
streams = sorted({r['Stream_Id'] for r in step}, key=lambda s: -sum(E(r) - S(r) for r in step if r['Stream_Id'] == s))
main = streams[0]
ev = []
for r in step:
    ev.append((S(r), 1, r['Stream_Id'])); ev.append((E(r), -1, r['Stream_Id']))
ev.sort()
busy = collections.Counter(); cur = collections.Counter(); last = t0
for t, d, s in ev:
    n = sum(1 for v in cur.values() if v > 0)
    key = (n, cur[main] > 0)
    busy[key] += t - last
    last = t
    cur[s] += d
print(f"step span {(t1 - t0) / 1e3:.0f} us; main stream = {main}")
for (n, m), v in sorted(busy.items()):
    print(f"  {n} stream(s) busy, main {'busy' if m else 'idle'}: {v / 1e3:8.0f} us")
mrows = [r for r in step if r['Stream_Id'] == main]
gaps = [(S(b) - E(a), name(a), name(b)) for a, b in zip(mrows, mrows[1:])]
print(f"main stream: {len(mrows)} kernels, busy {sum(E(r) - S(r) for r in mrows) / 1e3:.0f} us, gaps total {sum(g for g, _, _ in gaps) / 1e3:.0f} us "
      f"(median {sorted(g for g, _, _ in gaps)[len(gaps) // 2] / 1e3:.1f} us)")
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print(f"   gap {g / 1e3:7.1f} us  after {a}  before {b}")
# what the other streams were doing during the largest main-stream gap
gl = sorted(((S(b) - E(a), E(a), S(b)) for a, b in zip(mrows, mrows[1:])), reverse=True)[0]
print(f"largest gap [{(gl[1] - t0) / 1e3:.0f} us .. {(gl[2] - t0) / 1e3:.0f} us of the step]: kernels of the other streams inside it")
for r in step:
    if r['Stream_Id'] != main and E(r) > gl[1] and S(r) < gl[2]:
        print(f"     stream {r['Stream_Id']}  {(S(r) - gl[1]) / 1e3:8.1f} .. {(E(r) - gl[1]) / 1e3:8.1f} us  {name(r)}")
# overlap share per main-stream kernel family
others = sorted((S(r), E(r)) for r in step if r['Stream_Id'] != main)
def overlap(a, b):
    tot = 0
    for s, e in others:
        if e <= a: continue
        if s >= b: break
        tot += min(e, b) - max(s, a)
    return min(tot, b - a)
fam = collections.defaultdict(lambda: [0, 0, 0])
for r in mrows:
    f = fam[name(r)]; f[0] += 1; f[1] += E(r) - S(r); f[2] += overlap(S(r), E(r))
print("main-stream kernel families: launches, total us, share of run time with another stream active")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"   {k:62s} {v[0]:3d} {v[1] / 1e3:8.0f} us  {v[2] / max(v[1], 1):5.2f}")
