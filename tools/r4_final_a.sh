#!/bin/bash
# round 4, end-of-round call 1: the whole -m gpu suite, smoke(), the default bench line
set -o pipefail
R=gpurun_out/r4fa; rm -rf $R; mkdir -p $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x ${PYTEST_ARGS} > $R/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $R/tests.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $R/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $R/smoke.log
timeout -k 10 600 python bench.py > $R/bench.json 2> $R/bench.err; echo "bench rc=$?"; tail -c 2500 $R/bench.json
