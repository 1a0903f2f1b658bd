#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: time, MFMA utilisation, wave-cycle shares.
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x clock x 1024 SIMDs); SQ_WAVE_CYCLES-family counters are in
quad-cycles (MI355X_MICROARCH.md), so only their ratios are used.  Usage: pmc_summary.py <counter_collection.csv> [clock_GHz]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
clk = float(sys.argv[2]) if len(sys.argv) > 2 else 2.1
val = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
calls = collections.Counter()
seen = set()
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:70]
    val[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        calls[k] += 1
print("kernel,calls,total_ms,avg_us,mfma_util,active_share,wait_any_share,wait_inst_share,lds_bank_conflict_share")
for k in sorted(dur, key=lambda x: -dur[x]):
    v = val[k]
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    util = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (dur[k] * clk * 1024) if dur[k] else 0
    print(f"\"{k}\",{calls[k]},{dur[k]/1e6:.3f},{dur[k]/calls[k]/1e3:.1f},{util:.3f},{v.get('SQ_ACTIVE_INST_ANY',0)/wc:.3f},"
          f"{v.get('SQ_WAIT_ANY',0)/wc:.3f},{v.get('SQ_WAIT_INST_ANY',0)/wc:.3f},{v.get('SQ_LDS_BANK_CONFLICT',0)/wc:.4f}")
