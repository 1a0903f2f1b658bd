#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout -k 10 200 python3 tools/replay_debug.py C3 2 6 --chains 2>&1 | tail -3
timeout -k 10 200 python3 tools/replay_debug.py C0 8 6 2>&1 | tail -3
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3d/tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r3d/tests.log
timeout -k 10 300 python3 bench.py --host-inputs --steps 20 --warmup 5 > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.err; echo "bench rc=$?"; python3 -c "
import json
d=json.loads(open('gpurun_out/r3d/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','host_inputs','cpu_baseline')})
print(d['roofline'])"
