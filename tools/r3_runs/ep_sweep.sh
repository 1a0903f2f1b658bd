#!/bin/bash
export SWEEP_TILES=128x128
for lib in libmsc_hip.so libmsc_hip_ep2.so libmsc_hip_oldgemm.so; do
  echo "== $lib"
  MSC_LIB_PATH=multimodal_sc_amd/csrc/$lib python3 tools/gemm_sweep.py 2>&1 | grep -v amdgpu.ids | cut -c1-75
done
