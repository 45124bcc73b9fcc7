#!/bin/bash
# usage: tools/r2_prof.sh "<env assignments>" tag     -- kernel trace of 3 bench steps, per-kernel totals per step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
env $1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$2 -- python bench.py --steps 2 --warmup 1 --cpu-rhs 0 --skip-extras > gpurun_out/$2.json 2> gpurun_out/$2.err || { echo FAILED; tail -5 gpurun_out/$2.err; exit 1; }
python tools/r2_kstats.py gpurun_out/prof_$2 3
