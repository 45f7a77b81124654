#!/bin/bash
# round 3, step d: evidence for the march stack's rejection -- lane statistics (-DBT_LANESTAT build) and PMC counters of C4 with
# the march stack off and on (same library, bt_tuning through the CLI's environment knobs).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04d; mkdir -p $O
for mp in 0 128; do
  echo "== lanestat BT_END_GAME=16 BT_MARCH_POOL=$mp"
  BT_ONLY=volume,cornell2,scene BT_END_GAME=16 BT_MARCH_POOL=$mp BT_MARCH_ENTER=32 timeout -k 10 200 bash tools/run_with_lib.sh libbendy_hip_lanestat.so python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids | tee -a $O/lanestat_march_pool.log
done
echo "== lanestat BT_END_GAME=0 BT_MARCH_POOL=0"
BT_ONLY=volume,cornell2,scene BT_END_GAME=0 BT_MARCH_POOL=0 timeout -k 10 200 bash tools/run_with_lib.sh libbendy_hip_lanestat.so python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids | tee -a $O/lanestat_march_pool.log
for mp in 0 128; do
  BT_END_GAME=16 BT_MARCH_POOL=$mp BT_MARCH_ENTER=32 timeout -k 10 300 python3 tools/pmc_collect.py --workload C4 --calls 3 --passes sq,classes,waits --out $O/pmc_C4_march_pool_$mp.json > /dev/null 2>$O/pmc_$mp.err || tail -3 $O/pmc_$mp.err
  python3 - $O/pmc_C4_march_pool_$mp.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); x = d["derived"]
print(sys.argv[1], "kernel ms (profiler)", round(d["cli"]["kernel_ms_under_profiler"], 3), {k: round(x[k], 4) for k in ("valu_issue_frac", "lanes_active", "valu_lane_weighted_frac", "scalar_per_cu_cycle", "valu_per_wave", "salu_per_wave", "lds_per_wave", "wave_issuing_pct", "wave_issue_stalled_pct", "wave_waiting_pct") if k in x}, "VALU insts", int(d["mean_per_launch"]["SQ_INSTS_VALU"]))
PY
done
