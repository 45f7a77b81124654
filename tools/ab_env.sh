#!/bin/bash
# A/B of run-time knobs: each argument is "VAR=value" (or "none"), timed with tools/time_workloads.py
cd $GRAFT_REPO_ROOT
for kv in "$@"; do
  echo "=== $kv"
  if [ "$kv" = "none" ]; then python tools/time_workloads.py; else env $kv python tools/time_workloads.py; fi
done
