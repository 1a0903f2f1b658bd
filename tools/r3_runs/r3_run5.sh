#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
timeout -k 10 1000 python -m pytest tests/test_frames_gpu.py tests/test_hw_probe_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x > gpurun_out/r3e/tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r3e/tests.log
MSC_LIB_PATH=$GRAFT_REPO_ROOT/multimodal_sc_amd/csrc/libmsc_hip_np256.so SWEEP_TILES=128x128,256x128n timeout -k 10 400 python3 tools/gemm_sweep.py lin_ 2>&1 | grep -v wgrad | tee gpurun_out/r3e/sweep_np256_lin.txt | tail -10
MSC_LIB_PATH=$GRAFT_REPO_ROOT/multimodal_sc_amd/csrc/libmsc_hip_np256.so SWEEP_TILES=128x128,256x128n timeout -k 10 400 python3 tools/gemm_sweep.py conv_dgrad 2>&1 | tee gpurun_out/r3e/sweep_np256_convd.txt | tail -12
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3e/kt -o run -- python3 bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r3e/kt.log 2>&1
python3 tools/trace_step.py gpurun_out/r3e/kt/run_kernel_trace.csv 14 > gpurun_out/r3e/step_by_stream.txt; python3 tools/trace_gaps.py gpurun_out/r3e/kt/run_kernel_trace.csv > gpurun_out/r3e/step_gaps.txt; head -24 gpurun_out/r3e/step_gaps.txt; rm -f gpurun_out/r3e/kt/run_kernel_trace.csv
for R in 0 1; do MSC_REPLAY=$R timeout -k 10 200 python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('replay=$R', round(d['ms_per_step'],3), 'ms')"; done
