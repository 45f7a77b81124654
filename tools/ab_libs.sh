#!/bin/bash
# A/B on the GPU box: run tools/time_workloads.py against each variant library given as argument
# (file names under bendy_tracer_amd/; "libbendy_hip.so" means the freshly built one).
cd $GRAFT_REPO_ROOT
cp bendy_tracer_amd/libbendy_hip.so /tmp/base.so
for v in "$@"; do
  echo "=== $v"
  if [ "$v" = "libbendy_hip.so" ]; then cp /tmp/base.so bendy_tracer_amd/libbendy_hip.so; else cp bendy_tracer_amd/$v bendy_tracer_amd/libbendy_hip.so; fi
  python tools/time_workloads.py
done
cp /tmp/base.so bendy_tracer_amd/libbendy_hip.so
