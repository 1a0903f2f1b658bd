#!/bin/bash
for v in 512 1024 2048; do
  echo "MSC_WGRAD_SMALL_MAX=$v"; MSC_WGRAD_SMALL_MAX=$v python3 tools/call_table.py wgrad 2>&1 | grep -E "1, 1, 2, 0, 0, 1\)|stem_s2d_conv_wgrad"
done
b() { python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms')"; }
for r in 1 2 3; do
  MSC_WGRAD_SMALL_MAX=512 b "max=512"
  MSC_WGRAD_SMALL_MAX=1024 b "max=1024"
  MSC_WGRAD_SMALL=0 b "small=0"
done
