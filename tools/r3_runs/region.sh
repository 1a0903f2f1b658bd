#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/region; rm -rf $R; mkdir -p $R
rocprofv3 --kernel-trace --output-format csv -d $R/kt -o run -- python3 bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-roofline > $R/kt.log 2>&1
python3 tools/trace_region.py $R/kt/run_kernel_trace.csv meanpool_fwd 14 110 > $R/region.txt; rm -f $R/kt/run_kernel_trace.csv; cat $R/region.txt
