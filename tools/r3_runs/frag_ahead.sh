#!/bin/bash
# all fragment reads of a K step issued before its first MFMA (product) vs the compiler's one-read-ahead schedule (alt build fa0)
python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "linear or conv or gemm or stem or block" 2>&1 | tail -2
export SWEEP_TILES=128x128
for lib in libmsc_hip.so libmsc_hip_fa0.so; do
  echo "== $lib"; MSC_LIB_PATH=multimodal_sc_amd/csrc/$lib python3 tools/gemm_sweep.py 2>&1 | grep -v amdgpu | cut -c1-75
done
b() { python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms')"; }
for r in 1 2 3; do
  b "frag ahead"
  MSC_LIB_PATH=multimodal_sc_amd/csrc/libmsc_hip_fa0.so b "compiler schedule"
done
