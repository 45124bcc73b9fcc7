#!/bin/bash
# round 3: the blocked engine's level kernels and the supernodal engine's sn_gemm on the same factor, no stream overlap.
# HISTORICAL: ran at commit 84ad225's parent, before the supernodal engine (RSP_SUPERNODAL) was removed; its output is
# profiles/r03_blocked_vs_sn_kstats.txt.  Kept as the record of how that file was made; it no longer runs as is.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
set -o pipefail
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-rhs 0 --skip-extras > gpurun_out/r3_base.json 2> gpurun_out/r3_base.err || { echo BASE FAILED; tail -5 gpurun_out/r3_base.err; exit 1; }
cat gpurun_out/r3_base.json
export RSP_PAIR_OVERLAP=0
RSP_DUMP_LEVELS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof_blocked -- python bench.py --steps 2 --warmup 1 --cpu-rhs 0 --skip-extras > gpurun_out/r3_blocked.json 2> gpurun_out/r3_blocked.err || { echo BLOCKED FAILED; tail -5 gpurun_out/r3_blocked.err; exit 1; }
python tools/r2_kstats.py gpurun_out/r3_prof_blocked 3 > gpurun_out/r3_blocked_kstats.txt && cat gpurun_out/r3_blocked_kstats.txt
RSP_SUPERNODAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof_sn -- python bench.py --steps 2 --warmup 1 --cpu-rhs 0 --skip-extras > gpurun_out/r3_sn.json 2> gpurun_out/r3_sn.err || { echo SN FAILED; tail -5 gpurun_out/r3_sn.err; exit 1; }
python tools/r2_kstats.py gpurun_out/r3_prof_sn 3 > gpurun_out/r3_sn_kstats.txt && cat gpurun_out/r3_sn_kstats.txt
