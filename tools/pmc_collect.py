#!/usr/bin/env python3
"""Counter passes over the render kernel with rocprofv3, one `--pmc` group per run (gpurun refuses --pmc together with
API traces; FETCH_SIZE and WRITE_SIZE do not fit one pass -- MI355X_MICROARCH.md, "rocprofv3 PMC slots").

    python3 tools/pmc_collect.py --workload C3 [--calls 4] [--out profiles/<tag>/pmc_C3.json]

The profiled program is the headless CLI (bendy_tracer_amd/bendy-tracer-hip, C++ over the C ABI: no Python under the
profiler, a pass takes a few seconds), rendering `calls` launches of the workload's samples into an HBM-resident
frame.  Used by bench.py (live `roofline.traffic` / VALU figures in the bench line) and tools/profile.sh.

Derived figures (per launch of the render kernel, means over the launches):
  hbm_bytes           = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   -- both counters are in KiB; on gfx950 FETCH_SIZE tallies
                        128-B requests at 64 B, so it is doubled (the guide's HBM section; calibrated in r01c on a 33 MB frame)
  kernel_cycles       = GRBM_GUI_ACTIVE / 8                     -- the 8 XCDs count in parallel
  valu_per_simd_cycle = SQ_INSTS_VALU / (4 * CUs * kernel_cycles); the peak is 0.5 (a wave64 instruction issues over
                        2 cycles on a SIMD-32)
  lanes_active        = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
  issue_weighted      = sum over opcode classes (SQ_INSTS_VALU_<class> x issue cycles of the class, measured by
                        tools/valu_microbench.hip) / (4 * CUs * kernel_cycles): the share of VALU issue slots in use
"""
import argparse
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "bendy_tracer_amd", "bendy-tracer-hip")
WORKLOADS = {  # name: (scene, width, height, samples per launch)      -- bench.py's table
    "C3": ("scene", 1920, 1080, 64),
    "C2": ("cornell2", 512, 512, 16),
    "C4": ("volume", 1920, 1080, 64),
    "C5": ("scene", 3840, 2160, 256),
    "cornell1080": ("cornell", 1920, 1080, 64),
    "cloud1080": ("cloud", 1920, 1080, 64),
}
PASSES = {
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "sq": ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_SALU",
           "SQ_INSTS_SMEM", "GRBM_GUI_ACTIVE"],
    "classes": ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
                "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_LDS"],
    "waits": ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_SALU",
              "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM", "SQ_WAVE_CYCLES"],
}
N_SIMD = 1024          # MI355X: 256 CUs x 4 SIMD-32
VALU_PEAK = 0.5        # wave64 instructions per SIMD-cycle (MI355X_MICROARCH.md: 2 cycles each)
HBM_PEAK = 8.0e12      # bytes/s (spec; ~6.3e12 achievable)
FP32_PEAK = 157.3e12   # flop/s, vector FP32 (MI355X_MICROARCH.md): 256 CUs x 128 lanes x 2 (FMA) x 2.4 GHz
FP32_CLOCK_HZ = 2.4e9
# the guide's prices per wave64 instruction: 2 cycles plain, 4 for transcendental and half-rate integer (64-bit / mul_lo / mul_hi)
GUIDE_CLASS_COST = {"ADD_F32": 2.0, "MUL_F32": 2.0, "FMA_F32": 2.0, "TRANS_F32": 4.0, "INT32": 2.0, "INT64": 4.0, "CVT": 2.0,
                    "OTHER": 2.0}
# issue cycles per wave64 instruction by PMC class; replaced by profiles/valu_issue_costs.json when it exists
DEFAULT_CLASS_COST = {"ADD_F32": 2.0, "MUL_F32": 2.0, "FMA_F32": 2.0, "TRANS_F32": 4.0, "INT32": 2.0, "INT64": 8.0, "CVT": 2.0,
                      "OTHER": 2.0}


def source_hash():
    """sha256 over the kernel and launch sources: stamps a PMC summary with the code it was measured on (the GPU box has
    no .git, so the commit hash is not available there; tools/profile.sh adds it when run from a checkout)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "bendy_tracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".h", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def class_costs(key="class_cost"):
    try:
        return {**DEFAULT_CLASS_COST, **json.load(open(os.path.join(ROOT, "profiles", "valu_issue_costs.json")))[key]}
    except Exception:
        return dict(DEFAULT_CLASS_COST)


def cli_command(workload, calls, stats_path=None, shard=None, spp=None):
    """shard = (rank, world): the launch of ONE rank of a `world`-rank job (its interleaved tiles, bench.py --gpus N);
    spp overrides the workload's samples per launch (weak scaling: 64 * N)."""
    scene, w, h, spp0 = WORKLOADS[workload]
    spp = spp or spp0
    cmd = [CLI, "--output", "full", "--width", str(w), "--height", str(h), "--subsample", "1", "--samples", str(spp * calls),
           "--samples-per-call", str(spp), "--scene", os.path.join(ROOT, "scenes", f"{scene}.json.gz"), "--no-screenshot",
           "--quiet"]
    if stats_path:
        cmd += ["--stats-json", stats_path]
    if shard and shard[1] > 1:
        cmd += ["--shard", f"{shard[0]},{shard[1]}"]
    return cmd


def run_pass(counters, cmd, workdir, timeout=300):
    """One rocprofv3 --pmc run -> {counter: [value per render-kernel launch]}, kernel meta."""
    if shutil.which("rocprofv3") is None:
        raise RuntimeError("rocprofv3 not on PATH")
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", workdir, "-o", "pmc", "--", *cmd],
                       cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    files = glob.glob(os.path.join(workdir, "**", "*counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        raise RuntimeError("rocprofv3 --pmc %s failed (rc %d): %s" % (" ".join(counters), r.returncode,
                                                                      r.stdout.decode("utf-8", "replace")[-400:]))
    per_dispatch = collections.defaultdict(dict)
    meta = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if "bt_render_kernel" not in row["Kernel_Name"]:
                continue
            key = (row.get("Dispatch_Id"), row["Kernel_Name"])
            per_dispatch[key][row["Counter_Name"]] = per_dispatch[key].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            meta = {k: row[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                                        "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size") if k in row}
    out = collections.defaultdict(list)
    for _, vals in sorted(per_dispatch.items(), key=lambda kv: int(kv[0][0] or 0)):
        for k, v in vals.items():
            out[k].append(v)
    return out, meta


def collect(workload, calls=4, passes=("fetch", "write", "sq", "classes"), keep=None, shard=None, spp=None):
    """Runs the passes; -> dict with per-launch means, derived figures, kernel meta and the source hash."""
    tmp = keep or tempfile.mkdtemp(prefix="bt_pmc_")
    os.makedirs(tmp, exist_ok=True)
    stats_path = os.path.join(tmp, "cli_stats.json")
    means, meta, launches = {}, {}, {}
    for name in passes:
        vals, m = run_pass(PASSES[name], cli_command(workload, calls, stats_path, shard, spp), os.path.join(tmp, name))
        meta = m or meta
        for k, v in vals.items():
            v = v[1:] if len(v) > 1 else v            # the first launch pays scratch allocation / cold caches
            means[k] = sum(v) / len(v)
            launches[k] = len(v)
    res = {"workload": workload, "source_sha": source_hash(), "kernel": meta, "launches_averaged": launches,
           "mean_per_launch": means}
    if shard and shard[1] > 1:
        res["shard"] = {"rank": shard[0], "world": shard[1], "samples_per_launch": spp or WORKLOADS[workload][3]}
    try:
        st = json.load(open(stats_path))["calls"]
        st = st[1:] if len(st) > 1 else st
        res["cli"] = {"kernel_ms_under_profiler": sum(c["kernel_ms"] for c in st) / len(st),
                      "segments": st[-1]["segments"], "samples": st[-1]["samples"]}
    except Exception:
        pass
    res["derived"] = derive(means)
    d = res["derived"]
    if "fp32_flops" in d and res.get("cli", {}).get("kernel_ms_under_profiler") and "kernel_cycles" in d:
        # flop rate at the clock the counters ran at: flops per kernel cycle x the 2.4 GHz the peak is quoted at
        d["fp32_flop_frac"] = d["fp32_flops"] / d["kernel_cycles"] * FP32_CLOCK_HZ / FP32_PEAK
    if keep is None:
        shutil.rmtree(tmp, ignore_errors=True)
    return res


def derive(m):
    d = {}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        d["hbm_bytes"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
        d["hbm_read_bytes"] = 2.0 * m["FETCH_SIZE"] * 1024.0
        d["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024.0
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        d["kernel_cycles"] = cyc
        if "SQ_INSTS_VALU" in m:
            d["valu_per_simd_cycle"] = m["SQ_INSTS_VALU"] / (N_SIMD * cyc)
            d["valu_issue_frac"] = d["valu_per_simd_cycle"] / VALU_PEAK
        if "SQ_INSTS_SALU" in m:
            d["scalar_per_cu_cycle"] = (m["SQ_INSTS_SALU"] + m.get("SQ_INSTS_SMEM", 0.0)) / (N_SIMD / 4 * cyc)
        cls = {k[len("SQ_INSTS_VALU_"):]: v for k, v in m.items() if k.startswith("SQ_INSTS_VALU_")}
        if cls and "SQ_INSTS_VALU" in m:
            cost = class_costs()
            other = max(0.0, m["SQ_INSTS_VALU"] - sum(cls.values()))
            weighted = sum(v * cost.get(k, 2.0) for k, v in cls.items()) + other * cost["OTHER"]
            d["valu_class_counts"] = {**{k: round(v) for k, v in cls.items()}, "OTHER": round(other)}
            d["valu_issue_cycles_weighted"] = weighted
            d["valu_issue_weighted_frac"] = weighted / (N_SIMD * cyc)
            d["mean_issue_cycles_per_valu_inst"] = weighted / m["SQ_INSTS_VALU"]
            mixed_cost = class_costs("class_cost_mixed")        # mixed-stream opcode costs (tools/make_issue_costs.py)
            mixed = sum(v * mixed_cost.get(k, 2.0) for k, v in cls.items()) + other * mixed_cost["OTHER"]
            d["valu_issue_mixed_frac"] = mixed / (N_SIMD * cyc)
            guide = sum(v * GUIDE_CLASS_COST.get(k, 2.0) for k, v in cls.items()) + other * GUIDE_CLASS_COST["OTHER"]
            d["valu_issue_guide_frac"] = guide / (N_SIMD * cyc)
    if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m:
        d["lanes_active"] = m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"])
        if "valu_issue_frac" in d:
            d["valu_lane_weighted_frac"] = d["valu_issue_frac"] * d["lanes_active"]
        if all(("SQ_INSTS_VALU_" + k) in m for k in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32")):
            # FP32 operations actually performed: (ADD + MUL + 2 FMA + TRANS) wave instructions x 64 lanes x the share of lanes active
            d["fp32_flops"] = (m["SQ_INSTS_VALU_ADD_F32"] + m["SQ_INSTS_VALU_MUL_F32"] + 2.0 * m["SQ_INSTS_VALU_FMA_F32"] +
                               m["SQ_INSTS_VALU_TRANS_F32"]) * 64.0 * d["lanes_active"]
    if "SQ_WAVES" in m:
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH"):
            if k in m:
                d[k.lower().replace("sq_insts_", "") + "_per_wave"] = m[k] / m["SQ_WAVES"]
    if "SQ_WAVE_CYCLES" in m:
        for name, key in (("wave_issuing_pct", "SQ_ACTIVE_INST_ANY"), ("wave_issue_stalled_pct", "SQ_WAIT_INST_ANY"),
                          ("wave_waiting_pct", "SQ_WAIT_ANY")):
            if key in m:
                d[name] = 100.0 * m[key] / m["SQ_WAVE_CYCLES"]
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--calls", type=int, default=4)
    ap.add_argument("--passes", default="fetch,write,sq,classes,waits")
    ap.add_argument("--out")
    ap.add_argument("--shard", default=None, help="rank,world: one rank's launch of a sharded job")
    ap.add_argument("--spp", type=int, default=None, help="samples per launch (default: the workload's)")
    ap.add_argument("--commit", default=None, help="git commit hash to stamp into the summary (tools/profile.sh)")
    args = ap.parse_args()
    shard = tuple(int(x) for x in args.shard.split(",")) if args.shard else None
    res = collect(args.workload, args.calls, tuple(args.passes.split(",")), shard=shard, spp=args.spp)
    if args.commit:
        res["commit"] = args.commit
    text = json.dumps(res, indent=1)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    sys.exit(main())
