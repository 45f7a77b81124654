#!/bin/bash
# Packed launches: a guarded first render, launch-shape fuzz against the oracle, then timing of one block per workgroup (u) against
# packed without (p) and with (c) the compacting drain.  usage: tools/gpu_packed.sh <tag> [variant.so]
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout -k 5 90 python - > $O/first_render.log 2>&1 <<'PY'
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import numpy as np, torch
import bendy_tracer_amd as bendy, bt_oracle_py as oracle
from helpers import gpu_render, oracle_render
for name, w, h, spp in (("scene", 400, 260, 4), ("cornell2", 330, 200, 3), ("volume", 330, 200, 5), ("scene", 1920, 1080, 2)):
    print(name, w, h, spp, "...", flush=True)
    buf, st, _ = gpu_render(bendy, name, w, h, spp, tuning={"packed": 2})
    it, seg = oracle_render(oracle, name, w, h, spp, threads=16)
    print(name, "packed", st.packed, "segments", st.segments == seg, "bits", np.array_equal(buf.numpy(), it), "kernel ms", st.kernel_ms, flush=True)
PY
rc=$?; grep -v amdgpu.ids $O/first_render.log
if [ $rc -ne 0 ]; then echo "first render rc=$rc"; exit 1; fi
if grep -q False $O/first_render.log; then echo "MISMATCH"; exit 1; fi
timeout -k 10 300 python tools/fuzz_shapes.py 200 14 > $O/fuzz_packed.log 2>&1; rc=$?; tail -3 $O/fuzz_packed.log
if [ $rc -ne 0 ]; then exit 1; fi
for f in 512x512 768x512 1920x1080; do
  BT_FRAME=$f BT_MODES=u,p,c BT_T=1,2,4,8,16,32 timeout -k 10 200 python tools/time_shallow.py 2>&1 | grep -v amdgpu.ids | tee -a $O/time_packed.log
done
if [ -n "$2" ]; then bash tools/gpu_ab_variants.sh $1 $2; fi
