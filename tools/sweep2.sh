# usage: bash tools/sweep2.sh "<bench args>" tag ...
while [ $# -ge 2 ]; do
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-rhs 0 --skip-extras $1 > gpurun_out/$2.json 2> gpurun_out/$2.err && echo "$2: $(python tools/summ.py gpurun_out/$2.json 2>/dev/null | head -1)" || { echo "$2 failed"; tail -3 gpurun_out/$2.err; exit 1; }
  shift 2
done
