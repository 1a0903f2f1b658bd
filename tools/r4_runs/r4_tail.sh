#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r4tail; rm -rf $R; mkdir -p $R
rocprofv3 --kernel-trace --output-format csv -d $R/kt -o run -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-roofline --no-parity --alt-frames-per-gpu 0 > $R/kt.log 2>&1 || exit 1
python3 tools/trace_tail.py $R/kt/run_kernel_trace.csv > $R/tail.txt 2>&1; python3 tools/trace_region.py $R/kt/run_kernel_trace.csv adam_kernel 45 6 > $R/tail_region.txt
rm -f $R/kt/run_kernel_trace.csv
cut -c1-170 $R/tail_region.txt | tail -52
