#!/usr/bin/env python3
"""Developer tool: instruction-cache and LDS-conflict counters of the render kernel for one workload.
usage (GPU box): python3 tools/pmc_icache.py C3 [C4 ...]"""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_collect as pc

GROUPS = [["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQ_IFETCH", "SQ_WAVE_CYCLES"],
          ["SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES"],
          ["SQC_DCACHE_REQ", "SQC_DCACHE_HITS", "SQC_DCACHE_MISSES", "SQ_INSTS_SMEM", "SQ_WAVE_CYCLES"]]
for wl in sys.argv[1:] or ["C3"]:
    out = {}
    for g in GROUPS:
        tmp = tempfile.mkdtemp(prefix="bt_pmc_")
        try:
            vals, _ = pc.run_pass(g, pc.cli_command(wl, 2), tmp)
            out.update({k: sum(v) / len(v) for k, v in vals.items()})
        except Exception as e:      # a counter this build of rocprofv3 does not know
            out["error_" + g[0]] = str(e)[-200:]
    print(wl, json.dumps(out))
