#!/bin/bash
# round 3: GPU parity tests touching the wide items, then the bench (C2, C3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_gpu_round2.py -x -q --timeout 300 --timeout-method thread -m gpu -k "wide_items or switch_variant" > gpurun_out/r3_items_tests.txt 2>&1
rc=$?; tail -5 gpurun_out/r3_items_tests.txt; [ $rc -eq 0 ] || exit $rc
bash tools/r3_ab.sh "--steps 10 --warmup 2" "RSP_X=0" && bash tools/r3_ab.sh "--config c3 --steps 3 --warmup 1" "RSP_X=0"
