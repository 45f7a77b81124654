#!/bin/bash
# C2 (cornell2 512x512x16) under queue / slices / tiles-per-workgroup settings.  usage: tools/ab_c2.sh
cd $GRAFT_REPO_ROOT
export BT_ONLY=cornell2
for cfg in "" "BT_QUEUE=0" "BT_SLICES=1" "BT_SLICES=2" "BT_SLICES=4" "BT_SLICES=8" "BT_SLICES=1 BT_TILES_PER_WG=2" "BT_QUEUE=2" "BT_PHASE_VOTE=3" ""; do
  echo "=== ${cfg:-default}"
  env $cfg python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids
done
