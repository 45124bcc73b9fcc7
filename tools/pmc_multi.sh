# usage: bash tools/pmc_multi.sh tag "<counters pass 1>" "<counters pass 2>" ...   one rocprofv3 --pmc run per pass
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift; k=0
for pass in "$@"; do
  k=$((k+1))
  timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_${tag}_$k -- python bench.py --steps 2 --warmup 1 --cpu-rhs 0 --skip-extras > gpurun_out/pmc_${tag}_$k.json 2> gpurun_out/pmc_${tag}_$k.err || { echo "pass $k failed"; tail -3 gpurun_out/pmc_${tag}_$k.err; exit 1; }
  python tools/pmc_agg.py gpurun_out/pmc_${tag}_$k
done
