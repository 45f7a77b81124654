#!/bin/bash
# golden parity + time_workloads under BT_SLICES = auto, 2, 4, 8, 16, 32 with the current library.  usage: tools/ab_slices_now.sh
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gpu_matches_golden or random_scenes_bit_exact or every_slice_count or sliced" 2>&1 | tail -2
for s in - 2 4 8 16 32; do
  echo "=== BT_SLICES=$s"
  if [ "$s" = "-" ]; then python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids; else BT_SLICES=$s python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids; fi
done
