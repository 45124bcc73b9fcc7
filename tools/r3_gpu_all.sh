#!/bin/bash
# the whole GPU suite as the driver runs it, output to gpurun_out/r3_gpu_all.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -u -m pytest tests/ -x -q --timeout 300 --timeout-method thread -m gpu > gpurun_out/r3_gpu_all.txt 2>&1
rc=$?; tail -15 gpurun_out/r3_gpu_all.txt; exit $rc
