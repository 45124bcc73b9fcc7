#!/bin/bash
# round 3: kernel trace and the two HBM counter passes (separate runs) of the GPU numeric LU at C4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RSP_LU_TRACE=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3lu_trace -- python tools/r3_lu_profile.py > gpurun_out/r3lu_trace.json 2> gpurun_out/r3lu_trace.err || { echo TRACE FAILED; tail -5 gpurun_out/r3lu_trace.err; exit 1; }
echo trace done
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r3lu_rd -- python tools/r3_lu_profile.py > gpurun_out/r3lu_rd.json 2> gpurun_out/r3lu_rd.err || { echo RD FAILED; tail -5 gpurun_out/r3lu_rd.err; exit 1; }
echo rd done
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r3lu_wr -- python tools/r3_lu_profile.py > gpurun_out/r3lu_wr.json 2> gpurun_out/r3lu_wr.err || { echo WR FAILED; tail -5 gpurun_out/r3lu_wr.err; exit 1; }
echo wr done
