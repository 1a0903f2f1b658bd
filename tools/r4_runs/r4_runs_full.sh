#!/bin/bash
# the whole GPU test suite, one process
set -o pipefail
mkdir -p gpurun_out/r4g
timeout -k 10 1150 python -m pytest tests -q -m gpu -x > gpurun_out/r4g/gpu_tests.txt 2>&1; echo "gpu tests rc=$?" | tee -a gpurun_out/r4g/status.txt
tail -15 gpurun_out/r4g/gpu_tests.txt | cut -c1-300
