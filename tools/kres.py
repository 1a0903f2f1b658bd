#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` stderr.
Usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> res.txt ; tools/kres.py res.txt [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
names = [b.split('\n')[0].strip() for b in blocks]
dem = subprocess.run(['c++filt'], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for b, n in zip(blocks, dem):
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    n = n.replace('void ', '').replace('unsigned short', 'bf16').replace('(GemmParams)', '')
    if flt in n:
        sc, oc, ld = g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')
        print(f"{n[:70]:70s} vgpr {g('VGPRs'):4d} agpr {g('AGPRs'):4d} scratch {sc:4d} occ {oc} lds {ld:6d} sgpr {g('SGPRs')}")
