#!/bin/bash
# usage: tools/r2_sweep.sh "<env assignments>" ...   -- one bench.py run per argument, output under gpurun_out/sweep/
mkdir -p gpurun_out/sweep
i=0
for cfg in "$@"; do
  i=$((i+1))
  echo "=== $cfg" | tee -a gpurun_out/sweep/log.txt
  env $cfg timeout -k 10 400 python bench.py --steps 10 --warmup 2 --cpu-rhs 0 --skip-extras > gpurun_out/sweep/run_$i.json 2> gpurun_out/sweep/run_$i.err || { echo "FAILED rc=$?" | tee -a gpurun_out/sweep/log.txt; tail -5 gpurun_out/sweep/run_$i.err; exit 1; }
  python - <<PY | tee -a gpurun_out/sweep/log.txt
import json
d=json.loads(open("gpurun_out/sweep/run_$i.json").read().strip().splitlines()[-1])
print("ms_per_step %.3f frac %.4f launches %s/%s check %s setup %s"%(d["ms_per_step"],d["roofline"]["frac"],d["config"]["lsolve_launches"],d["config"]["ltsolve_launches"],d["check"],{k:round(v,2) for k,v in d["setup"].items()}))
PY
done
