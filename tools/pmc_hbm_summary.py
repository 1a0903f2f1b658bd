#!/usr/bin/env python3
"""HBM bytes per kernel from two rocprofv3 --pmc passes (FETCH_SIZE in one, WRITE_SIZE in the other; both in KB).
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled.
Usage: pmc_hbm_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [summary.csv]
With a summary path the table goes there and <summary.csv>.meta.json records the sha256 of csrc/gemm.hip the counters were
taken on, kernel and launch planning (bench.py reports roofline.traffic only while that still matches the source it runs)."""
import collections, csv, hashlib, json, os, sys


def load(path, counter):
    tot, dur, calls = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:70]
        tot[k] += float(r["Counter_Value"]) * 1024.0
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        calls[k] += 1
    return tot, dur, calls


f, fd, fc = load(sys.argv[1], "FETCH_SIZE")
w, wd, wc = load(sys.argv[2], "WRITE_SIZE")
out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
print("kernel,calls,total_ms,fetch_bytes_corrected_x2,write_bytes,bytes_per_launch,GB_per_s", file=out)
for k in sorted(fd, key=lambda x: -fd[x]):
    fb, wb = 2.0 * f[k], w.get(k, 0.0)
    n = fc[k]
    print(f"\"{k}\",{n},{fd[k] / 1e6:.3f},{fb:.0f},{wb:.0f},{(fb + wb) / n:.0f},{(fb + wb) / fd[k]:.0f}", file=out)
if len(sys.argv) > 3:
    out.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("gemm.hip", "gemm_pp.hip", "gemm_skinny.hip", "conv_c64.hip", "conv_c64w.hip", "conv_stem.hip", "gemm_plan.h"):       # the same list as bench.gemm_source_sha
        h.update(open(os.path.join(root, "multimodal_sc_amd", "csrc", f), "rb").read())
    sha = h.hexdigest()[:16]
    # steps the profiled command ran (warm-up + timed): PMC_STEPS, default 4 = `bench.py --steps 3 --warmup 1` of tools/round_profile.sh
    json.dump({"gemm_hip_sha256_16": sha, "fetch_csv": os.path.basename(sys.argv[1]), "write_csv": os.path.basename(sys.argv[2]),
               "steps_profiled": int(os.environ.get("PMC_STEPS", "4"))}, open(sys.argv[3] + ".meta.json", "w"))
