#!/bin/bash
# Packed launches: correctness against the oracle, then the build against a variant library (e.g. the previous commit's) on the
# BASELINE workloads and small launches.  usage: tools/gpu_packed.sh <tag> [variant.so]
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 300 python tools/fuzz_shapes.py 150 13 > $O/fuzz_packed.log 2>&1; rc=$?; tail -3 $O/fuzz_packed.log
if [ $rc -ne 0 ]; then exit 1; fi
if [ -n "$2" ]; then bash tools/gpu_ab_variants.sh $1 $2; fi
