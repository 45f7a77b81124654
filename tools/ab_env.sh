#!/bin/bash
# A/B on the GPU box: tools/time_workloads.py under each value of an environment knob.
# usage: ab_env.sh VAR value1 value2 ...   ("-" = unset)
cd $GRAFT_REPO_ROOT
var=$1; shift
for v in "$@"; do
  echo "=== $var=$v"
  if [ "$v" = "-" ]; then env -u $var python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids; else env $var=$v python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids; fi
done
