#!/bin/bash
# quick SQ counter profile of one bench workload: tools/pmc_quick.sh <tag> <workload>
export TMPDIR=/tmp
TAG=$1; WL=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$TAG
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
BENCH="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --workload $WL"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $OUT/a -o pmc -- $BENCH > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o pmc -- $BENCH > $OUT/b.log 2>&1
python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for g in 'ab':
    for r in csv.DictReader(open('$OUT/%s/pmc_counter_collection.csv'%g)):
        if 'bt_render' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
m={k:sum(v)/len(v) for k,v in agg.items()}
for k in sorted(m): print(f'{k:28s} {m[k]:.4g}')
cyc=m['GRBM_GUI_ACTIVE']/8
print('VALUBusy %.1f%%'%(100*m['SQ_ACTIVE_INST_VALU']*4/1024/cyc))
print('VALUUtilization %.1f%%'%(100*m['SQ_THREAD_CYCLES_VALU']/(m['SQ_ACTIVE_INST_VALU']*64)))
wc=m['SQ_WAVE_CYCLES']
print('wave time: active %.1f%% wait_inst %.1f%% wait_any %.1f%%'%(100*m['SQ_ACTIVE_INST_ANY']/wc,100*m['SQ_WAIT_INST_ANY']/wc,100*m['SQ_WAIT_ANY']/wc))
print('kernel cycles %.4g  VALU inst/wave %.0f SMEM/wave %.0f SALU/wave %.0f'%(cyc, m['SQ_INSTS_VALU']/m['SQ_WAVES'], m['SQ_INSTS_SMEM']/m['SQ_WAVES'], m['SQ_INSTS_SALU']/m['SQ_WAVES']))
PY
