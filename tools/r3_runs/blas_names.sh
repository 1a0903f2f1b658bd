#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/blas_kt -o run -- python3 $GRAFT_REPO_ROOT/tools/probes/blas_probe.py > $GRAFT_REPO_ROOT/gpurun_out/blas_kt.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/blas_kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:30]:
    print(r["Calls"], r["AverageNs"], r["Name"][:400])
PY
