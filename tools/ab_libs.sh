#!/bin/bash
# A/B: run tools/time_workloads.py against each variant library given as argument.
cd $GRAFT_REPO_ROOT
cp bendy_tracer_amd/libbendy_hip.so /tmp/base.so
for v in "$@"; do
  echo "=== $v"
  cp bendy_tracer_amd/$v bendy_tracer_amd/libbendy_hip.so
  python tools/time_workloads.py
done
cp /tmp/base.so bendy_tracer_amd/libbendy_hip.so
