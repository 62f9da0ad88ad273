#!/bin/bash
# round 6: the wide path (three-qubit tomography) -- its tests, then where its time goes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6j; mkdir -p $O
timeout 1500 python -m pytest tests/test_wide_tomography.py -q -m gpu -x > $O/pytest_wide.log 2>&1; tail -40 $O/pytest_wide.log
timeout 600 python tools/wide_bench.py ${N:-1e6} > $O/wide_bench.txt 2>&1; tail -12 $O/wide_bench.txt
timeout 300 python tools/plugin_time.py > $O/plugin_time.txt 2>&1; tail -5 $O/plugin_time.txt
