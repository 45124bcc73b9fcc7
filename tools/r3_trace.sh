#!/bin/bash
# usage: tools/r3_trace.sh <tag> "<bench args>" [ENV=VAL ...]   -- kernel trace of 2 bench steps, launch timeline of the last step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; args=$2; shift; shift
for kv in "$@"; do export "$kv"; done
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof_$tag -- python bench.py $args --steps 2 --warmup 1 --cpu-rhs 0 --skip-extras > gpurun_out/r3_$tag.json 2> gpurun_out/r3_$tag.err || { echo FAILED; tail -5 gpurun_out/r3_$tag.err; exit 1; }
python tools/r3_timeline.py gpurun_out/r3_prof_$tag > gpurun_out/r3_timeline_$tag.txt; cut -c1-6000 gpurun_out/r3_timeline_$tag.txt
