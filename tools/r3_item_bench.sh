#!/bin/bash
# round 3 microbenchmark sweep of the wide-item kernel (tools/item_bench.hip, built beforehand into build/)
cd $GRAFT_REPO_ROOT
B=./build/item_bench
for cfg in "2048 256 8 256" "2048 64 8 256" "144 992 4 256" "356 448 4 256" "2048 256 8 128"; do
  for v in $VARIANTS; do
    for part in 0 40; do
      timeout -k 5 60 $B $cfg $v ${MT:-4} $part || exit 1
    done
  done
done
