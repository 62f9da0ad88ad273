#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r5f
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5f/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r5f/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
