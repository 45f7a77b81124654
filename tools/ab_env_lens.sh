#!/bin/bash
# A/B on the GPU box: tools/lens_time.py under each value of an environment knob.  usage: ab_env_lens.sh VAR v1 v2 ... ("-" = unset)
cd $GRAFT_REPO_ROOT
var=$1; shift
for v in "$@"; do
  echo "=== $var=$v"
  if [ "$v" = "-" ]; then env -u $var python tools/lens_time.py 2>&1 | grep -v amdgpu.ids; else env $var=$v python tools/lens_time.py 2>&1 | grep -v amdgpu.ids; fi
done
