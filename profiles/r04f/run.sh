#!/bin/bash
# round 3, step f: single loop, end-game compaction in the rect builds only.  Tests, timings, then the full check.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04f; mkdir -p $O
timeout -k 10 150 python -m pytest tests -m gpu -x -q -k "path_pool or shallow_launches or golden" > $O/pytest_pool.log 2>&1; rc=$?; echo "pool tests rc=$rc"; tail -3 $O/pytest_pool.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
echo "== default"; timeout -k 10 120 python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids | tee $O/time_default.log
for eg in 0 16 32 48 64; do
  echo "== BT_END_GAME=$eg"; BT_ONLY=cornell2,cornell BT_END_GAME=$eg timeout -k 10 120 python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_end_game_rects.log
done
for f in 768x512 512x512; do
  echo "== shallow $f"; BT_MODES=auto BT_FRAME=$f timeout -k 10 200 python tools/time_shallow.py 2>&1 | grep -v amdgpu.ids | tee $O/time_shallow_$f.log
done
