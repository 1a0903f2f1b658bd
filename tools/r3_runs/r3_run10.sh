#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
MSC_LIB_PATH=$GRAFT_REPO_ROOT/multimodal_sc_amd/csrc/libmsc_hip_dephase.so timeout -k 10 400 python3 tools/gemm_dephase.py 0,2,4,6,8,12 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3j/dephase.txt
