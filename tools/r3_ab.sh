#!/bin/bash
# usage: tools/r3_ab.sh "<bench args>" "ENV=VAL ..." "ENV=VAL ..." ...   -- the bench once per environment set, ms per step of each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args=$1; shift
i=0
for envs in "$@"; do
  i=$((i+1))
  ( for kv in $envs; do export "$kv"; done; timeout -k 10 900 python bench.py $args --cpu-rhs 0 --skip-extras > gpurun_out/r3_ab_$i.json 2> gpurun_out/r3_ab_$i.err ) || { echo "FAILED: $envs"; tail -5 gpurun_out/r3_ab_$i.err; exit 1; }
  python - "$envs" gpurun_out/r3_ab_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(f"{sys.argv[1]:60s} ms/step {d['ms_per_step']:.3f}  frac {d['roofline']['frac']:.4f}  residual {d['check'].get('residual_max_all_columns')}")
PY
done
