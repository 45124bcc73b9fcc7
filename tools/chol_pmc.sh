# usage: bash tools/chol_pmc.sh   -- kernel trace and the two counter passes of the numeric Cholesky at C2 (tools/chol_prof.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/chol_trace -- python tools/chol_prof.py 1024 > gpurun_out/chol_trace.txt 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/chol_fetch -- python tools/chol_prof.py 1024 > gpurun_out/chol_fetch.txt 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/chol_write -- python tools/chol_prof.py 1024 > gpurun_out/chol_write.txt 2>&1 &&
echo collected
