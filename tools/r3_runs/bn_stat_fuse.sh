#!/bin/bash
# BatchNorm-backward statistics in the producing dgrad epilogue (bn1 of the 8 residual blocks) vs a reduction kernel per BatchNorm
python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "conv or bn or block or stem" 2>&1 | tail -3
python3 -m pytest tests/test_model_gpu.py -q -m gpu -x 2>&1 | tail -3
b() { python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms')"; }
for r in 1 2 3; do
  MSC_BN_STAT_FUSE=1 b "fused=1"
  MSC_BN_STAT_FUSE=0 b "fused=0"
done
