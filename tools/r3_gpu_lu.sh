#!/bin/bash
# round 3: the GPU LU tests alone
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_gpu_lu.py -x -q -m gpu --timeout 300 --timeout-method thread 2>&1 | tee gpurun_out/r3_gpu_lu.log | tail -25
