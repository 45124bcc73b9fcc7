#!/bin/bash
# round 3: selected GPU tests (-k expression in $1), output tail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/ -x -q --timeout 300 --timeout-method thread -m gpu -k "$1" > gpurun_out/r3_tests.txt 2>&1
rc=$?; tail -25 gpurun_out/r3_tests.txt; exit $rc
