#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
for rep in 1 2; do for O in recorded rr main; do MSC_REPLAY_ORDER=$O timeout -k 10 200 python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('order=$O', round(d['ms_per_step'],3), 'ms')"; done; done | tee gpurun_out/r3f/order_ab.txt
for O in rr main; do MSC_REPLAY_ORDER=$O timeout -k 10 300 python3 tools/replay_debug.py C3 2 6 --chains 2>&1 | tail -1; done
MSC_REPLAY_ORDER=rr timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3f/kt -o run -- python3 bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r3f/kt.log 2>&1
python3 tools/trace_gaps.py gpurun_out/r3f/kt/run_kernel_trace.csv > gpurun_out/r3f/step_gaps_rr.txt; head -14 gpurun_out/r3f/step_gaps_rr.txt; rm -f gpurun_out/r3f/kt/run_kernel_trace.csv
