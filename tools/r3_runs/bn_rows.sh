#!/bin/bash
# BatchNorm-backward reduction: rows per row lane (8 product, 4, 32 = the old 128 rows per block for every layer)
python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "bn or conv or block or stem" 2>&1 | tail -2
for v in "" _rpl4 _rpl32; do
  echo "lib$v:"; MSC_LIB_PATH=multimodal_sc_amd/csrc/libmsc_hip$v.so python3 tools/call_table.py msc_bn_bwd 2>&1 | grep "msc_bn_bwd " | cut -c1-110
done
b() { python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms')"; }
for r in 1 2 3; do
  b "8 rows per lane"
  MSC_LIB_PATH=multimodal_sc_amd/csrc/libmsc_hip_rpl4.so b "4 rows per lane"
  MSC_LIB_PATH=multimodal_sc_amd/csrc/libmsc_hip_rpl32.so b "128 rows per block (old)"
done
