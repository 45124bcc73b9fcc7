#!/bin/bash
# round 2's wrong one-column block under rocprofv3, with the hipMemcpy2D form of the copy back in place
# (RSP_COPY_2D_CHECK=1: the block is read back and compared right after the copy) -- plain, then under a kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RSP_COPY_2D_CHECK=1
echo "== plain"
timeout -k 10 400 python bench.py --steps 1 --warmup 1 --cpu-rhs 0 > gpurun_out/r3_c2d_plain.json 2> gpurun_out/r3_c2d_plain.err || { echo PLAIN FAILED; tail -3 gpurun_out/r3_c2d_plain.err; }
grep -a "copy check" gpurun_out/r3_c2d_plain.err | sort | uniq -c | tail -5
python3 -c "
import json; e=json.loads(open('gpurun_out/r3_c2d_plain.json').read().strip().splitlines()[-1])['extras']; print({k:e[k] for k in ('residual_1rhs','x_error_1rhs','cholsol_one_shot_x_error','cholsol_cached_x_error')})"
echo "== rocprofv3 --kernel-trace"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3_c2d_trace -- python bench.py --steps 1 --warmup 1 --cpu-rhs 0 > gpurun_out/r3_c2d_trace.json 2> gpurun_out/r3_c2d_trace.err || { echo TRACE FAILED; tail -3 gpurun_out/r3_c2d_trace.err; }
grep -a "copy check" gpurun_out/r3_c2d_trace.err | sort | uniq -c | tail -5
python3 -c "
import json; e=json.loads(open('gpurun_out/r3_c2d_trace.json').read().strip().splitlines()[-1])['extras']; print({k:e[k] for k in ('residual_1rhs','x_error_1rhs','cholsol_one_shot_x_error','cholsol_cached_x_error')})"
exit 0
