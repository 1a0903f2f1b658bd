#!/bin/bash
# fusion-layer GELU: derivative stored by fc1's forward epilogue (default) vs evaluated in fc2's dgrad epilogue (MSC_GELU_DSAVE=0)
python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "linear" 2>&1 | tail -3
b() { python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms')"; }
for r in 1 2 3; do
  MSC_GELU_DSAVE=1 b "dsave=1"
  MSC_GELU_DSAVE=0 b "dsave=0"
done
