#!/bin/bash
# kernel trace of the codec / head region of one step (between the last fusion layer's forward and the first one's backward)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r4h2; rm -rf $R; mkdir -p $R
rocprofv3 --kernel-trace --output-format csv -d $R/kt -o run -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-roofline --no-parity --alt-frames-per-gpu 0 > $R/kt.log 2>&1 || exit 1
python3 tools/trace_region.py $R/kt/run_kernel_trace.csv recon_loss_kernel 30 45 > $R/head_region.txt
rm -f $R/kt/run_kernel_trace.csv
cut -c1-170 $R/head_region.txt
