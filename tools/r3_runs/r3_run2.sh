#!/bin/bash
# round 3, second GPU call: the replay path — focused tests, host enqueue time and step time, eager loop vs launch replay
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py tests/test_bench_gpu.py -m gpu -q -x -k "replay or hipgraph or dp or bench or C4 or c4 or checkpoint or clip or determinism or f32_end_to_end_parity" > gpurun_out/r3b/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3b/tests.log
for B in 64 32 16; do for R in 0 1; do MSC_REPLAY=$R timeout -k 10 200 python3 tools/host_overhead.py $B 2>&1 | tail -1; done; done | tee gpurun_out/r3b/host_overhead.txt
for B in 64 32; do for R in 0 1 0 1; do MSC_REPLAY=$R timeout -k 10 200 python3 bench.py --batch $B --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B replay=$R', round(d['ms_per_step'],3), 'ms', round(d['value']), 'frames/s')"; done; done | tee gpurun_out/r3b/bench_ab.txt
