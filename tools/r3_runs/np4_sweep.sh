#!/bin/bash
# 256x128 tile with FOUR waves (128x64 wave tiles, two workgroups per CU) against the product tiles, dense shapes of the path
export SWEEP_TILES=128x128,256x128n
MSC_LIB_PATH=multimodal_sc_amd/csrc/libmsc_hip_np4.so python3 tools/gemm_sweep.py lin_ 2>&1 | grep -v amdgpu | cut -c1-150
MSC_LIB_PATH=multimodal_sc_amd/csrc/libmsc_hip_np4.so SWEEP_B=32 python3 tools/gemm_sweep.py lin_ 2>&1 | grep -v amdgpu | cut -c1-150
