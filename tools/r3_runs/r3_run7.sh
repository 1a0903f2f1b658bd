#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g; rm -f gpurun_out/r3g/program.txt
MSC_REPLAY_DUMP=gpurun_out/r3g/program.txt timeout -k 10 200 python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
wc -l gpurun_out/r3g/program.txt
