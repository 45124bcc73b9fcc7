#!/bin/bash
cd $GRAFT_REPO_ROOT
B=./build/item_bench
for span in 100000 1000000; do
for cfg in "232 992 4 256 4 4 20" "232 992 4 256 1 4 20" "1712 144 2 256 1 2 10" "856 144 2 256 1 4 10" "450 448 8 256 2 4 20"; do
  timeout -k 5 60 $B $cfg $span || exit 1
done; done
