# usage: bash tools/prof2.sh "<bench args>" tag nlaunches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$2 -- python bench.py --steps 2 --warmup 1 --cpu-rhs 0 --skip-extras $1 > gpurun_out/$2.json 2> gpurun_out/$2.err
python tools/levels.py gpurun_out/prof_$2 $3 | tail -1
