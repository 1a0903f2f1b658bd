#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
timeout -k 10 300 python3 tools/replay_debug.py C3 2 6 --chains 2>&1 | tail -4
timeout -k 10 300 python3 tools/replay_debug.py C3 2 6 2>&1 | tail -4
timeout -k 10 300 python3 tools/replay_debug.py C1 3 6 2>&1 | tail -4
timeout -k 10 300 python3 tools/replay_debug.py C0 8 6 2>&1 | tail -4
timeout -k 10 400 python -m pytest tests/test_frames_gpu.py -m gpu -q -x 2>&1 | tail -15
SWEEP_TILES=128x128,128x128p,256x128,256x256 timeout -k 10 600 python3 tools/gemm_sweep.py lin_ 2>&1 | tee gpurun_out/r3c/sweep_lin.txt | tail -14
SWEEP_TILES=128x128,128x128p,128x64,64x128 timeout -k 10 600 python3 tools/gemm_sweep.py conv_fwd 2>&1 | tee gpurun_out/r3c/sweep_convf.txt | tail -14
SWEEP_TILES=128x128,128x128p,128x64,64x128 timeout -k 10 600 python3 tools/gemm_sweep.py conv_dgrad 2>&1 | tee gpurun_out/r3c/sweep_convd.txt | tail -14
