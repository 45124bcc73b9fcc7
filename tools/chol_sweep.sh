# usage: bash tools/chol_sweep.sh "<env>" tag ...   -- numeric Cholesky time at C2 (1024^2) and 3-D 48^3 under an environment
while [ $# -ge 2 ]; do
  a=$(env $1 timeout -k 10 200 python tools/chol_prof.py 2>/dev/null | tail -1) && b=$(env $1 timeout -k 10 200 python tools/chol_prof.py 3d48 2>/dev/null | tail -1) && echo "$2: C2 $a | 3d48 $b" || { echo "$2 failed"; exit 1; }
  shift 2
done
