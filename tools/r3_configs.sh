#!/bin/bash
# round 3: the other BASELINE.json configurations through bench.py's --config modes (and tools/configs.py for C4), plus the
# single-GPU width sweep at C3 that the projected strong-scaling curve is made from
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in 256 128 64 32; do
  timeout -k 10 400 python bench.py --config c3 --nrhs $w --steps 3 --warmup 1 --cpu-rhs 0 --skip-extras > gpurun_out/r3_c3_w$w.json 2> gpurun_out/r3_c3_w$w.err || { echo "C3 $w FAILED"; tail -3 gpurun_out/r3_c3_w$w.err; exit 1; }
  echo "c3 width $w done"
done
timeout -k 10 400 python bench.py --config c5 --steps 2 --warmup 1 --cpu-rhs 0 > gpurun_out/r3_c5.json 2> gpurun_out/r3_c5.err || { echo C5 FAILED; tail -3 gpurun_out/r3_c5.err; exit 1; }
echo "c5 done"
timeout -k 10 400 python tools/configs.py c4 > gpurun_out/r3_c4.json 2> gpurun_out/r3_c4.err || { echo C4 FAILED; tail -3 gpurun_out/r3_c4.err; exit 1; }
echo "c4 done"
