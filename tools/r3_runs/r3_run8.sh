#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
prof() { # tag, env...
  tag=$1; shift

  ( export "$@"; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3h/$tag -o run -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r3h/$tag.log 2>&1 )
  python3 tools/trace_gaps.py gpurun_out/r3h/$tag/run_kernel_trace.csv > gpurun_out/r3h/gaps_$tag.txt; echo "=== $tag"; head -40 gpurun_out/r3h/gaps_$tag.txt | grep -v "^   gap  *[0-9]\.\|^   gap  *1[0-9]\." ; rm -f gpurun_out/r3h/$tag/run_kernel_trace.csv
}
prof base MSC_X=0
prof noside MSC_SIDE_STREAM=0
prof nobranch MSC_BRANCH_STREAMS=0
prof mainprio MSC_MAIN_PRIO=1
for rep in 1 2; do for P in "" 1; do MSC_MAIN_PRIO=$P timeout -k 10 200 python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('main_prio=[$P]', round(d['ms_per_step'],3), 'ms')"; done; done
