#!/bin/bash
# step time and isolated attention time against the wave counts of the two backward passes (dQ / dK,dV)
b() { python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms')"; }
for cfg in "8 8" "4 4" "8 4" "4 8" "8 8" "4 4" "8 4" "4 8"; do
  set -- $cfg
  export MSC_ATT_DQ_WAVES=$1 MSC_ATT_BWD_WAVES=$2
  echo "dq=$1 dkv=$2: $(python3 tools/attn_bench.py 2>&1 | tail -1)"
  b "dq=$1 dkv=$2 step"
done
