#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3o
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "attention or invariants or two_chains or determinism" > gpurun_out/r3o/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3o/tests.log
for rep in 1 2; do for K in 0 1; do echo -n "kvlds=$K "; MSC_ATT_DKV_KVLDS=$K timeout -k 10 200 python3 tools/attn_bench.py 2>&1 | tail -1; done; done | tee gpurun_out/r3o/attn_ab.txt
b() { python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms')"; }
for rep in 1 2; do for K in 0 1; do MSC_ATT_DKV_KVLDS=$K b "kvlds=$K"; done; done | tee -a gpurun_out/r3o/attn_ab.txt
