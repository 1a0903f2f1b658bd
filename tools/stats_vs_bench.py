#!/usr/bin/env python3
"""Does the committed rocprofv3 summary agree with bench.py's roofline figure?
    rocprofv3 --kernel-trace --stats ... -- python3 bench.py --instrumented-only N      (N + 1 single-stream passes: one warm-up)
usage: stats_vs_bench.py <run_kernel_stats.csv> <passes incl. warm-up> <bench json line file>
Sums the contraction kernel's rows (gemm_kernel<...> and the split-K reduce / finish kernels its C-ABI calls launch) per pass and
prints them beside kernel_ms_per_pass of the bench line."""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
passes = int(sys.argv[2])
gk = lambda n: any(k in n for k in ("gemm_kernel", "gemm_pp_kernel", "gemm_skinny_kernel", "conv_c64_kernel", "conv_c64_wgrad_kernel", "stem_wgrad_kernel"))
fam = lambda n: gk(n) or ("splitk_reduce_kernel" in n) or ("splitk_finish_kernel" in n)
tot = sum(float(r["TotalDurationNs"]) for r in rows if fam(r["Name"]))
calls = sum(int(r["Calls"]) for r in rows if fam(r["Name"]))
gem = sum(float(r["TotalDurationNs"]) for r in rows if gk(r["Name"]))
gcalls = sum(int(r["Calls"]) for r in rows if gk(r["Name"]))
line = [l for l in open(sys.argv[3]).read().splitlines() if l.startswith("{")][-1]
b = json.loads(line)
ms = b.get("kernel_ms_per_pass_median") or b["roofline"]["kernel_ms_per_step"]
gf = b.get("algorithmic_gflop_per_pass") or b["roofline"]["algorithmic_gflop_per_step"]
print(f"rocprofv3 stats: contraction family {tot / passes / 1e6:.3f} ms per pass ({calls // passes} kernel launches; gemm_kernel alone "
      f"{gem / passes / 1e6:.3f} ms, {gcalls // passes} launches, average {gem / gcalls / 1e3:.1f} us)")
print(f"bench.py (HIP events around the C-ABI calls): {ms:.3f} ms per pass -> {gf / ms:.1f} TFLOP/s; from the rocprof sum: {gf / (tot / passes / 1e6):.1f} TFLOP/s "
      f"(ratio {ms / (tot / passes / 1e6):.3f})")
