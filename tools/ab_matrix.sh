#!/bin/bash
# A/B matrix: libraries x env knobs.  usage: ab_matrix.sh "lib1 lib2" "none VAR=v ..."
cd $GRAFT_REPO_ROOT
cp bendy_tracer_amd/libbendy_hip.so /tmp/base.so
for v in $1; do
  if [ "$v" = "libbendy_hip.so" ]; then cp /tmp/base.so bendy_tracer_amd/libbendy_hip.so; else cp bendy_tracer_amd/$v bendy_tracer_amd/libbendy_hip.so; fi
  for kv in $2; do
    echo "=== $v $kv"
    if [ "$kv" = "none" ]; then python tools/time_workloads.py; else env $kv python tools/time_workloads.py; fi
  done
done
cp /tmp/base.so bendy_tracer_amd/libbendy_hip.so
