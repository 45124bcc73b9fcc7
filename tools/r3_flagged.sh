#!/bin/bash
# RSP_FLAGGED_LEVELS=1 (one right-hand side: runs of latency-bound levels as one launch each): parity tests, then the single-RHS figures
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
RSP_FLAGGED_LEVELS=1 timeout -k 10 700 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q --timeout 120 --timeout-method thread -m gpu > gpurun_out/r3_flag_tests.txt 2>&1
rc=$?; tail -4 gpurun_out/r3_flag_tests.txt
[ $rc -eq 0 ] || exit $rc
for f in 0 1; do
  RSP_FLAGGED_LEVELS=$f RSP_TIMING=1 timeout -k 10 400 python bench.py --steps 2 --warmup 1 --cpu-rhs 0 > gpurun_out/r3_flag_$f.json 2> gpurun_out/r3_flag_$f.err || exit 1
  grep -a "flagged levels" gpurun_out/r3_flag_$f.err | sort | uniq -c | head -4
  python3 -c "
import json; l=json.loads(open('gpurun_out/r3_flag_$f.json').read().strip().splitlines()[-1]); e=l['extras']; print('flagged=$f', 'ms_per_step', round(l['ms_per_step'],3), '1rhs_ms', round(e['cholsol_solve_1rhs_ms'],3), 'residual_1rhs', e['residual_1rhs'], 'x_err', e['x_error_1rhs'], 'cached_s', round(e['cholsol_cached_s'],4))"
done
