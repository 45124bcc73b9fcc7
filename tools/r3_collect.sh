#!/bin/bash
# round 3: the rocprofv3 evidence of the default bench command: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in passes of
# their own (a counter run carries no trace options).  Summaries: python tools/collect_profiles.py r03_c2_k256 ... afterwards.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r3c_trace gpurun_out/r3c_fetch gpurun_out/r3c_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3c_trace -- python bench.py --steps 2 --warmup 1 --cpu-rhs 0 --skip-extras > gpurun_out/r3c_trace.json 2> gpurun_out/r3c_trace.err || { echo TRACE FAILED; tail -5 gpurun_out/r3c_trace.err; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r3c_fetch -- python bench.py --steps 2 --warmup 1 --cpu-rhs 0 --skip-extras > gpurun_out/r3c_fetch.json 2> gpurun_out/r3c_fetch.err || { echo FETCH FAILED; tail -5 gpurun_out/r3c_fetch.err; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r3c_write -- python bench.py --steps 2 --warmup 1 --cpu-rhs 0 --skip-extras > gpurun_out/r3c_write.json 2> gpurun_out/r3c_write.err || { echo WRITE FAILED; tail -5 gpurun_out/r3c_write.err; exit 1; }
echo collected
