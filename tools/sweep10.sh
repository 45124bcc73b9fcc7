# usage: bash tools/sweep10.sh "<env>" tag ...   -- like sweep.sh, 10 timed steps
while [ $# -ge 2 ]; do
  env $1 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --cpu-rhs 0 --skip-extras > gpurun_out/$2.json 2> gpurun_out/$2.err && echo "$2: $(python tools/summ.py gpurun_out/$2.json 2>/dev/null | head -1)" || { echo "$2 failed"; tail -3 gpurun_out/$2.err; exit 1; }
  shift 2
done
