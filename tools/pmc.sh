# usage: bash tools/pmc.sh <counter> <tag>   -- one counter block per run (separate passes for FETCH_SIZE and WRITE_SIZE)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 240 rocprofv3 --pmc $1 --output-format csv -d gpurun_out/pmc_$2 -- python bench.py --steps 2 --warmup 1 --cpu-rhs 0 --skip-extras > gpurun_out/pmc_$2.json 2> gpurun_out/pmc_$2.err
ls gpurun_out/pmc_$2/*/ | head
