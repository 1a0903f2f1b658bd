#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "relu_ref or premasked or basic_block or conv_bn or f32_end_to_end or determinism or replay or bf16" > gpurun_out/r3k/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3k/tests.log
b() { python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms')"; }
for rep in 1 2 3; do for P in 0 1; do MSC_BLOCK_PREMASK=$P b "premask=$P"; done; done | tee gpurun_out/r3k/ab.txt
