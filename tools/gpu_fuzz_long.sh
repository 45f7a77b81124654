#!/bin/bash
# Long fuzz on the GPU box: 100 000 random scenes in four chunks (progress lines keep the run alive).  usage: tools/gpu_fuzz_long.sh <tag>
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-fuzzlong}; mkdir -p $O
for first in ${FUZZ_FIRST:-100000 125000 150000 175000}; do
  timeout -k 10 400 python tools/fuzz_sweep.py $first 25000 > $O/fuzz_sweep_$first.log 2>&1; rc=$?; echo "sweep from $first rc=$rc"; tail -1 $O/fuzz_sweep_$first.log
  if [ $rc -ne 0 ]; then exit 1; fi
done
