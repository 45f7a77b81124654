#!/bin/bash
# round 3, step g: clean kernel (no pool).  Tests, micro-diet A/B (skip spheres behind the origin; min/max range test of the
# sphere normal), interactive latency with the counter ring, PMC of C3.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04g; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
for v in libbendy_hip.so libbendy_hip_behind.so libbendy_hip_minmax.so libbendy_hip_both.so libbendy_hip.so; do
  echo "== $v"; timeout -k 10 150 bash tools/run_with_lib.sh $v python tools/time_workloads.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_micro_diet.log
done
echo "== golden + fuzz parity of the 'both' variant"; timeout -k 10 300 bash tools/run_with_lib.sh libbendy_hip_both.so python -m pytest tests -m gpu -x -q -k "golden or random_scenes or c3_scene or c4_volume" 2>&1 | tail -2 | tee $O/parity_both.log
python3 tools/pmc_collect.py --workload C3 --calls 4 --passes sq,classes --out $O/pmc_C3.json > /dev/null 2>$O/pmc.err || tail -3 $O/pmc.err
python3 - $O/pmc_C3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); x = d["derived"]
print("C3 default build: kernel ms (profiler)", round(d["cli"]["kernel_ms_under_profiler"], 3), {k: round(x[k], 4) for k in ("valu_issue_frac", "valu_issue_guide_frac", "lanes_active", "valu_lane_weighted_frac", "scalar_per_cu_cycle", "valu_per_wave", "salu_per_wave") if k in x}, "VALU insts", int(d["mean_per_launch"]["SQ_INSTS_VALU"]), x.get("valu_class_counts"))
PY
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > $O/bench_nopmc.log 2>$O/bench.err; echo "bench rc=$?"
python3 - $O/bench_nopmc.log <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("C3", d["value"], "ms/step", d["ms_per_step"], {k: (v["kernel_ms"], v.get("ms_per_render") or v.get("ms_per_call_synchronised")) for k, v in d["other_configs"].items()})
print(d["other_configs"]["interactive_default"])
PY
