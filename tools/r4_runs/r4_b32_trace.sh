#!/bin/bash
# round 4: kernel stats of the step at 32 frames per GPU (BASELINE configs[3] geometry) beside 64 frames, same box: which kernels do not halve
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r4b32; rm -rf $R; mkdir -p $R
for B in 32 64; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt$B -o run -- python3 bench.py --batch $B --steps 12 --warmup 5 --no-cpu-baseline --no-roofline --no-parity --alt-frames-per-gpu 0 > $R/kt$B.log 2>&1 || exit 1
  python3 tools/trace_step.py $R/kt$B/run_kernel_trace.csv 14 > $R/step_by_stream_$B.txt
  cp $R/kt$B/run_kernel_stats.csv $R/kernel_stats_$B.csv; rm -f $R/kt$B/run_kernel_trace.csv
done
python3 - <<'PY'
import csv, re
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("(anonymous namespace)::", "").replace("unsigned short", "bf16")
        c, t = int(r["Calls"]), float(r["TotalDurationNs"])
        a = d.setdefault(n, [0, 0.0]); a[0] += c; a[1] += t
    return d
a, b = load("gpurun_out/r4b32/kernel_stats_32.csv"), load("gpurun_out/r4b32/kernel_stats_64.csv")
rows = []
for n in set(a) | set(b):
    ca, ta = a.get(n, [0, 0.0]); cb, tb = b.get(n, [0, 0.0])
    rows.append((max(ta, tb), n, ca, ta, cb, tb))
rows.sort(reverse=True)
print(f"{'kernel':70s} calls32  avg32us | calls64  avg64us | total32/total64")
for _, n, ca, ta, cb, tb in rows[:45]:
    print(f"{n[:70]:70s} {ca:6d} {ta/max(ca,1)/1e3:8.1f} | {cb:6d} {tb/max(cb,1)/1e3:8.1f} | {ta/max(tb,1):.2f}")
print("sum of kernel time  32: %.2f ms   64: %.2f ms  (all steps of the run)" % (sum(v[1] for v in a.values())/1e6, sum(v[1] for v in b.values())/1e6))
PY
