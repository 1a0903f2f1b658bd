#!/bin/bash
python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "layernorm or ln_ or fusion" 2>&1 | tail -2
for rep in 1 2; do
for r in 1 2 4 8; do
  lib=multimodal_sc_amd/csrc/libmsc_hip_ln$r.so; [ $r = 4 ] && lib=multimodal_sc_amd/csrc/libmsc_hip.so
  echo "rows/wave=$r: $(MSC_LIB_PATH=$lib python3 tools/ln_bench.py 2>&1 | tail -1)"
done; done
