#!/bin/bash
cd $GRAFT_REPO_ROOT
cp bendy_tracer_amd/libbendy_hip.so /tmp/base.so
for v in "$@"; do
  echo "=== $v"
  if [ "$v" = "libbendy_hip.so" ]; then cp /tmp/base.so bendy_tracer_amd/libbendy_hip.so; else cp bendy_tracer_amd/$v bendy_tracer_amd/libbendy_hip.so; fi
  python tools/lens_time.py
done
cp /tmp/base.so bendy_tracer_amd/libbendy_hip.so
