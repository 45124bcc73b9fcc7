# usage: bash tools/single_sweep.sh "<env>" tag ...   -- single right-hand-side cholsol solve phase at C2 under an environment
while [ $# -ge 2 ]; do
  env $1 timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-rhs 0 > gpurun_out/$2.json 2> gpurun_out/$2.err && echo "$2: $(python -c "import json,sys; d=json.loads(open('gpurun_out/$2.json').read().strip().splitlines()[-1]); print('1rhs ms', round(d['extras']['cholsol_solve_1rhs_ms'],3), 'step ms', round(d['ms_per_step'],3))")" || { echo "$2 failed"; exit 1; }
  shift 2
done
