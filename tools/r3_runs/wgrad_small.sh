#!/bin/bash
# shorter / more K slices for weight gradients whose output is 1-4 tiles (1x1 downsample convolutions, stem)
python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "conv or stem or linear" 2>&1 | tail -2
for v in 1 0; do
  echo "MSC_WGRAD_SMALL=$v"; MSC_WGRAD_SMALL=$v python3 tools/call_table.py wgrad 2>&1 | grep -E "1, 1, 2, 0, 0, 1\)|stem_s2d_conv_wgrad|isolated sum"
done
b() { python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms')"; }
for r in 1 2 3; do
  MSC_WGRAD_SMALL=1 b "small=1"
  MSC_WGRAD_SMALL=0 b "small=0"
done
