#!/bin/bash
# C2 under the drain cadences: PMC lanes and kernel time.  usage: tools/gpu_c2_drain.sh <tag> variant.so
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for v in libbendy_hip.so $2 libbendy_hip.so $2; do
  echo "== $v" | tee -a $O/c2_drain.log
  BT_ONLY=cornell2 timeout -k 10 100 bash tools/run_with_lib.sh $v python tools/time_c3.py 60 2>&1 | grep -v "amdgpu.ids\|same file" | tee -a $O/c2_drain.log
done
for v in libbendy_hip.so $2; do
  cp bendy_tracer_amd/libbendy_hip.so /tmp/base.so; cp bendy_tracer_amd/$v bendy_tracer_amd/libbendy_hip.so
  python3 tools/pmc_collect.py --workload C2 --passes sq --out $O/pmc_C2_$v.json > /dev/null 2>$O/pmc_$v.err
  cp /tmp/base.so bendy_tracer_amd/libbendy_hip.so
  python3 -c "
import json; d=json.load(open('$O/pmc_C2_$v.json'))['derived']; print('$v', 'lanes', round(d['lanes_active'],4), 'valu/simd-cycle', round(d['valu_per_simd_cycle'],4), 'cycles', d['kernel_cycles'])" | tee -a $O/c2_drain.log
done
