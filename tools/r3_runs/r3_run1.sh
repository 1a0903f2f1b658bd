#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite, one bench line, and a kernel trace whose kernel-name census shows every
# kernel of a step that is NOT one of libmsc_hip.so's (torch add / fill kernels the launch replay must not depend on)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout -k 10 800 python -m pytest tests -m gpu -q -x > gpurun_out/r3a/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3a/tests.log
timeout -k 10 300 python3 bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r3a/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3a/kt -o run -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r3a/kt.log 2>&1
python3 - <<'PY'
import csv, collections
c = collections.Counter()
for r in csv.DictReader(open("gpurun_out/r3a/kt/run_kernel_trace.csv")):
    c[r["Kernel_Name"][:110]] += 1
with open("gpurun_out/r3a/kernel_census.txt", "w") as f:
    for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
        f.write(f"{v:6d} {k}\n")
print(sum(c.values()), "launches,", len(c), "distinct kernels")
PY
grep -v -E "gemm_kernel|_kernel<|msc_|splitk|kernel\(" gpurun_out/r3a/kernel_census.txt | head -20
rm -f gpurun_out/r3a/kt/run_kernel_trace.csv
