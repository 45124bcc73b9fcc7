#!/bin/bash
# the hipMemcpy2D check plain, under a kernel trace and under a counter pass (each a run of its own)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B=/tmp/r3_memcpy2d_check.bin
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/r3_memcpy2d_check.hip -o $B || exit 1
echo "== plain"; timeout -k 10 120 $B > gpurun_out/r3_m2d_plain.txt 2>&1; echo "rc $?"; tail -2 gpurun_out/r3_m2d_plain.txt
echo "== rocprofv3 --kernel-trace"; timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3_m2d_trace -- $B > gpurun_out/r3_m2d_trace.txt 2>&1; echo "rc $?"; grep -a "hipMemcpy2D:\|MISMATCH" gpurun_out/r3_m2d_trace.txt | tail -4
echo "== rocprofv3 --pmc FETCH_SIZE"; timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r3_m2d_pmc -- $B > gpurun_out/r3_m2d_pmc.txt 2>&1; echo "rc $?"; grep -a "hipMemcpy2D:\|MISMATCH" gpurun_out/r3_m2d_pmc.txt | tail -4
exit 0
