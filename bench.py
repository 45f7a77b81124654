#!/usr/bin/env python3
"""bench.py -- headline benchmark: Msamples/s of Tracer::render on the GPU.

    python bench.py --gpus N --steps K --warmup W [--workload C3] [--scaling weak|strong] [--backend nccl|rccl-abi|gloo]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for N > 1: typed from a bare shell (no WORLD_SIZE in the environment) the first one starts the second as a
CHILD process -- one rank per GPU, rendezvous on 127.0.0.1 at a free port -- relays rank 0's JSON line and exits with the
child's return code (no exec: the parent has not touched the GPU and never does).  `--print-launch` prints that command
as JSON instead of running it.

Workload (BASELINE.json metric "Msamples/s (pixels x spp / s) at 1080p", config C3): scene.json.gz, 1920x1080,
Subsample::None, Config = main.rs values.  One "step" is one Tracer::render call that adds `samples` rays per pixel
to a frame that stays resident in HBM (the reference's progressive pattern, main.rs:245-254; step i uses
sample_base = i * samples).

N > 1: the 16x16 pixel tiles are dealt round-robin to the ranks, each rank renders its tiles into a rank-local shard
of running sums, one RCCL all-gather (over xGMI) collects the shards and an un-permute kernel rebuilds the row-major
frame.  Shard sums are rank-local, so step i+1's render does not depend on step i's exchange: the all-gather +
un-permute of step i run on a second HIP stream underneath the render of step i+1.  Everything, including the last
exchange, is inside the timed region.
  --scaling weak   (default) samples per step = 64 * N: fixed work per GPU.
  --scaling strong samples per step stay what the workload names (so `--workload C5 --gpus 8 --scaling strong` IS
                   BASELINE configs[4]: 3840x2160x256 spp sharded over 8 GPUs); a second timed pass gathers only once,
                   after the last step ("the final framebuffer"), and is reported as `final_gather_only`.
  --backend nccl   torch.distributed's RCCL; rccl-abi = the library's own bt_comm_* entry points (the exchange a
                   non-Python host would call); gloo = rehearsal on a box with fewer GPUs than ranks.

N > 1 lines carry what a reader needs to check them: `ranks_joined` (the process group's size after init), `devices` (one
entry per rank: device index, name, PCI bus id, host pid), `verified_vs_single_rank` (by default one extra, untimed step is
rendered sharded + exchanged and compared bit for bit with the same step rendered by one rank alone; --no-verify skips it,
--verify-all compares the whole accumulated run instead) and a `roofline` measured on rank 0's own shard launch.

One JSON line on rank 0.  `roofline`: the kernel is VALU-bound, so `achieved` / `peak` are wave64 VALU
instructions per SIMD-cycle (peak 0.5: MI355X_MICROARCH.md, 2 cycles per instruction on a SIMD-32), with the HBM
bytes (`traffic`) and every counter MEASURED IN THIS RUN by rocprofv3 --pmc passes over the same workload
(tools/pmc_collect.py; separate passes, FETCH_SIZE doubled per the guide) -- unless rocprofv3 is unavailable, then the
committed summary profiles/pmc_live.json is quoted and flagged `stale` when its source hash differs from this tree.
`cpu_baseline` (N = 1): the CPU oracle -- a C port of the reference algorithm, NOT the Rust binary -- timed on the host
cores on a bounded sample of the same workload.

`other_configs` (N = 1 only; not part of `value`): a few renders each of BASELINE.json's other configurations on this GPU
-- C2, C4, C5 rendered whole on one GPU, the 1080p Cornell box -- with wall and kernel time per `Tracer::render`; C2 and C4
also carry their own PMC-derived fractions (`roofline`), measured in this run like C3's.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
# the host driver of this pool only supports dmabuf IPC: without it RCCL fails with hipIpcGetMemHandle: invalid argument
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

WORKLOADS = {
    # name: (scene, width, height, samples per step at N=1)
    "C3": ("scene", 1920, 1080, 64),
    "C2": ("cornell2", 512, 512, 16),
    "C4": ("volume", 1920, 1080, 64),
    "C5": ("scene", 3840, 2160, 256),
    "cornell1080": ("cornell", 1920, 1080, 64),
    "cloud1080": ("cloud", 1920, 1080, 64),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
VALU_PEAK = 0.5                # wave64 VALU instructions per SIMD-cycle (2 cycles each on a SIMD-32)
BYTES_PER_SEGMENT = 128        # 64-byte SoA ray state read + written once per segment
BYTES_PER_PIXEL = 16           # RGBA32F written once per pixel per render
SEED = 0x5EED
PMC_CACHE = os.path.join(ROOT, "profiles", "pmc_live.json")


def cpu_baseline(scene_name, w, h, budget_s=15.0):
    """Times the CPU oracle (test infrastructure) on a bounded sample of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bt_oracle_py as o
    sc = o.Scene.load(os.path.join(ROOT, "scenes", f"{scene_name}.json.gz"))
    cam = sc.find_by_tag("camera")
    sc.set_camera_aspect(cam, w / h)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))

    def run(spp, chunks, base=0):
        cfg = o.default_config(samples=spp, recursive=1, chunks=chunks, sample_base=base)
        t = time.perf_counter()
        o.render(sc, cam, cfg, w, h, SEED, nthreads=cores)
        return time.perf_counter() - t

    t1 = run(1, (32, 32))                                  # probe: 1 spp
    spp = int(max(1, min(64, 0.25 * budget_s / max(t1, 1e-3))))
    # "best CPU": 1024 dynamic tiles; repeat passes of `spp` samples (consecutive sample ranges, as the progressive
    # loop of main.rs does) until ~2/3 of the budget is spent, so the sample is 10-30 s of CPU work on any host
    t_fine, passes = 0.0, 0
    while t_fine < 0.66 * budget_s:
        t_fine += run(spp, (32, 32), base=passes * spp)
        passes += 1
    t_ref = run(spp, (8, 4))                               # reference-shaped: 8x4 tiles (main.rs:225-230)
    n = w * h * spp
    return {
        "value": round(n * passes / t_fine / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": f"{scene_name}.json.gz {w}x{h}: {passes} passes of {spp} spp (C oracle, recursive form, gcc -O3 "
                  f"-ffp-contract=off, {cores} threads, 32x32 dynamic tiles; {t_fine:.1f} s)",
        "reference_tiling_8x4_value": round(n / t_ref / 1e6, 3),
    }


def parity_figure(b, torch, scene_name, w=240, h=135, spp=16):
    """The second half of BASELINE.json's metric, "max |delta pixel| vs CPU ref": a small frame of the same scene rendered
    by the HIP path and by the CPU oracle (checker only).  Mean-framebuffer difference per channel against the oracle's
    recursive form (products nested as the reference nests them), and bit-identity with its iterative form."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bt_oracle_py as o
    path = os.path.join(ROOT, "scenes", f"{scene_name}.json.gz")
    sc = b.Scene.load(path)
    cam = sc.find_by_tag("camera")
    sc.set_camera_aspect(cam, w / h)
    buf = b.Buffer.new(w, h)
    b.Tracer.with_config(b.Config(chunks_x=8, chunks_y=4)).render(sc, cam, b.RenderConfig.with_samples(spp), buf, seed=SEED)
    torch.cuda.synchronize()
    got = buf.numpy()
    osc = o.Scene.load(path)
    ocam = osc.find_by_tag("camera")
    osc.set_camera_aspect(ocam, w / h)
    rec, _, _ = o.render(osc, ocam, o.default_config(samples=spp, recursive=1), w, h, SEED, nthreads=8)
    it, _, _ = o.render(osc, ocam, o.default_config(samples=spp, recursive=0), w, h, SEED, nthreads=8)
    return {"frame": f"{scene_name}.json.gz {w}x{h}x{spp}spp", "tolerance": 1e-4,
            "max_abs_delta_mean_vs_cpu_recursive_form": float(np.abs(got[..., :3] - rec[..., :3]).max() / spp),
            "bit_identical_to_cpu_iterative_form": bool(np.array_equal(got, it)),
            "note": "CPU ref = this repo's C restatement of the reference algorithm (the Rust binary cannot be built here and "
                    "seeds from OS entropy); whole frames at the BASELINE sizes are compared in tests/test_gpu_parity.py"}


def measure_pmc(workload, shard=None, spp=None, calls=4):
    """rocprofv3 --pmc passes over this workload, run as child processes BEFORE this process touches the GPU
    (tools/pmc_collect.py; the profiled program is the C++ CLI over the same library).  shard = (rank, world): one rank's
    launch of a sharded job.  Falls back to the committed summary, flagged with whether it was taken on this source tree."""
    import pmc_collect
    try:
        res = pmc_collect.collect(workload, calls=calls, passes=("fetch", "write", "sq", "classes"), shard=shard, spp=spp)
        res["measured"] = "in this run (rocprofv3 --pmc, separate passes: FETCH_SIZE | WRITE_SIZE | SQ | VALU classes)"
        res["stale"] = False
        return res
    except Exception as e:                                     # no rocprofv3, no counters for this user, ...
        why = str(e)[:300]
    try:
        key = workload if not shard or shard[1] == 1 else f"{workload}_shard{shard[1]}"
        res = json.load(open(PMC_CACHE))[key]
        res["measured"] = f"profiles/pmc_live.json (live measurement unavailable: {why})"
        res["stale"] = res.get("source_sha") != pmc_collect.source_hash()
        return res
    except Exception:
        return {"measured": f"unavailable: {why}", "stale": None, "mean_per_launch": {}, "derived": {}}


def cached_pmc(workload):
    """--no-pmc: the committed summary, flagged stale when its source hash differs from this tree."""
    try:
        import pmc_collect
        pmc = json.load(open(PMC_CACHE))[workload]
        pmc["measured"] = "profiles/pmc_live.json (--no-pmc)"
        pmc["stale"] = pmc.get("source_sha") != pmc_collect.source_hash()
        return pmc
    except Exception:
        return None


def pmc_fractions(pmc, k_ms):
    """The PMC-derived part of a `roofline` object (the same keys for every configuration).  Three prices for the VALU
    issue slots in use: `frac` = every wave64 instruction at 2 cycles (the guide's plain rate, a lower bound);
    `issue_cost_guide_frac` = the guide's price list (2 plain, 4 transcendental and half-rate integer);
    `issue_cost_weighted_frac` = this repo's own microbenchmark (profiles/valu_issue_costs.json), the upper end.
    `fp32_flop_frac` = FP32 operations actually performed (ADD + MUL + 2 FMA + TRANS, active lanes only) over the 157.3 TF/s
    vector peak."""
    r = {}
    if not pmc or not pmc.get("derived"):
        return r
    d, m = pmc["derived"], pmc.get("mean_per_launch", {})
    rnd = lambda k: round(d[k], 4) if k in d else None
    r["achieved"] = rnd("valu_per_simd_cycle")
    r["frac"] = rnd("valu_issue_frac")
    r["lane_weighted_frac"] = rnd("valu_lane_weighted_frac")
    r["issue_cost_guide_frac"] = rnd("valu_issue_guide_frac")
    r["issue_cost_weighted_frac"] = rnd("valu_issue_mixed_frac")
    r["issue_cost_pure_stream_bound"] = rnd("valu_issue_weighted_frac")
    r["fp32_flop_frac"] = rnd("fp32_flop_frac")
    r["lanes_active_per_valu_inst"] = rnd("lanes_active")
    r["scalar_insts_per_cu_cycle"] = rnd("scalar_per_cu_cycle")
    if "hbm_bytes" in d:
        r["traffic"] = int(d["hbm_bytes"])
        r["hbm_measured_frac"] = round(d["hbm_bytes"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    r["valu_wave_insts_per_launch"] = int(m["SQ_INSTS_VALU"]) if "SQ_INSTS_VALU" in m else None
    r["valu_class_counts"] = d.get("valu_class_counts")
    r["pmc"] = {"measured": pmc.get("measured"), "source_sha": pmc.get("source_sha"), "stale": pmc.get("stale"),
                "commit": pmc.get("commit"), "kernel_cycles": round(d["kernel_cycles"]) if "kernel_cycles" in d else None,
                "kernel_ms_under_profiler": pmc.get("cli", {}).get("kernel_ms_under_profiler")}
    if pmc.get("shard"):
        r["pmc"]["shard"] = pmc["shard"]
    return r


def lens_extension_rate(b, torch, scene_name, w, h, spp=64, steps=2):
    """Throughput of the same frame with the gravitational-lens EXTENSION switched on.  Not part of `value`:
    the reference has no lens code, so this mode has no reference behaviour and no parity claim
    beyond GPU == this repo's CPU oracle; BASELINE.json's configs[2] reads "scene.json.gz 1920x1080 64spp with
    gravitational-lens geodesic stepping", hence this number on exactly that frame and sample count."""
    lens = dict(centre=(0.6, 0.4, 4.0), rs=0.15, step=0.1, radius=6.0, max_steps=800)
    sc = b.Scene.load(os.path.join(ROOT, "scenes", f"{scene_name}.json.gz"))
    cam = sc.find_by_tag("camera")
    sc.set_camera_aspect(cam, w / h)
    sc.set_lens(**lens)
    buf = b.Buffer.new(w, h)
    tr = b.Tracer.with_config(b.Config(chunks_x=8, chunks_y=4))
    ms = []
    for i in range(steps + 1):
        tr.render(sc, cam, b.RenderConfig.with_samples(spp), buf, seed=SEED, sample_base=i * spp)
        st = sc.last_stats()
        ms.append(st.kernel_ms)
    k = statistics.mean(ms[1:])
    return {"value": round(w * h * spp / k / 1e3, 1), "unit": "Msamples/s", "kernel_ms": round(k, 3), "spp_per_step": spp,
            "rk4_steps_per_sample": round(st.lens_steps / st.samples, 1), "lens": lens,
            "note": "extension, not in the reference; fixed-step RK4 on the Schwarzschild null geodesic"}


def other_configs(b, torch, steps=3, pmc_by_config=None):
    """The other BASELINE.json configurations on this GPU, a few renders each (they are parity-test cases, not the
    headline; `value` stays C3): per `Tracer::render` the wall time around the call (synchronised) and the kernel time by
    the library's HIP events.  C5 is configs[4]'s frame and depth rendered whole on ONE GPU (its 8-GPU form needs the
    driver's node)."""
    out = {}
    for name in ("C2", "C4", "C5", "cornell1080"):
        scene_name, w, h, spp = WORKLOADS[name]
        sc = b.Scene.load(os.path.join(ROOT, "scenes", f"{scene_name}.json.gz"))
        cam = sc.find_by_tag("camera")
        sc.set_camera_aspect(cam, w / h)
        buf = b.Buffer.new(w, h)
        tr = b.Tracer.with_config(b.Config(chunks_x=8, chunks_y=4))
        wall, kern = [], []
        n = steps * 8 if w * h * spp < (1 << 24) else steps          # small configurations: more renders, first one untimed
        for i in range(n + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr.render(sc, cam, b.RenderConfig.with_samples(spp), buf, seed=SEED, sample_base=i * spp)
            torch.cuda.synchronize()
            wall.append((time.perf_counter() - t0) * 1e3)
            kern.append(sc.last_stats().kernel_ms)
        st = sc.last_stats()
        ms = statistics.mean(wall[1:])
        k_ms = statistics.mean(kern[1:])
        out[name] = {"workload": f"{scene_name}.json.gz {w}x{h}x{spp}spp", "value": round(w * h * spp / ms / 1e3, 1),
                     "unit": "Msamples/s", "ms_per_render": round(ms, 4), "kernel_ms": round(k_ms, 4),
                     "segments_per_sample": round(st.segments / st.samples, 4), "launches": st.launches,
                     "slices": st.slices, "packed": bool(st.packed), "scratch_bytes": st.scratch_bytes}
        if pmc_by_config and pmc_by_config.get(name):
            out[name]["roofline"] = {"bound": "valu", "peak": VALU_PEAK, **pmc_fractions(pmc_by_config[name], k_ms)}
        del buf, sc
    # the reference's interactive loop with its CLI defaults (main.rs:52-65, 245-254): a 768 x 512 window, one
    # Tracer::render per displayed frame with samples = 1 and Subsample::Subpixel(2), scene.json
    w, h = 768, 512
    sc = b.Scene.load(os.path.join(ROOT, "scenes", "scene.json.gz"))
    cam = sc.find_by_tag("camera")
    sc.set_camera_aspect(cam, w / h)
    buf = b.Buffer.new(w, h)
    tr = b.Tracer.with_config(b.Config(chunks_x=8, chunks_y=4))
    rc = b.RenderConfig.with_samples_subsample(1, b.Subsample(2))
    for _ in range(4):
        tr.render(sc, cam, rc, buf, seed=SEED)
    torch.cuda.synchronize()
    lat = []
    for _ in range(32):                              # the loop reads a preview back after every call: synchronised latency
        t0 = time.perf_counter()
        tr.render(sc, cam, rc, buf, seed=SEED)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    for _ in range(64):
        tr.render(sc, cam, rc, buf, seed=SEED)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 64
    out["interactive_default"] = {"workload": "scene.json.gz 768x512, 1 sample x Subpixel(2) per call (the reference CLI's defaults)",
                                  "value": round(w * h * 4 / ms / 1e3, 1), "unit": "Msamples/s", "ms_per_call_pipelined": round(ms, 4),
                                  "ms_per_call_synchronised": round(statistics.median(lat), 4),
                                  "kernel_ms": round(sc.last_stats().kernel_ms, 4), "slices": sc.last_stats().slices,
                                  "packed": bool(sc.last_stats().packed)}
    return out


class ShardExchange:
    """Rank-local shard of running sums + the frame exchange of the N > 1 path."""

    def __init__(self, b, torch, dist, w, h, rank, world, backend, overlap):
        self.b, self.torch, self.dist = b, torch, dist
        self.w, self.h, self.rank, self.world, self.backend, self.overlap = w, h, rank, world, backend, overlap
        self.shard = b.new_shard(w, h, world)
        self.gathered = torch.empty(world * self.shard.numel(), dtype=torch.float32, device="cuda")
        self.frame = b.Buffer.new(w, h)
        self.comm = None
        if backend == "rccl-abi":
            # the library's own communicator (include/bendy_hip.h bt_comm_*): rank 0's unique id reaches the others
            # through the process group that exists anyway for the barrier
            uid = [b.Comm.unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(uid, src=0)
            self.comm = b.Comm(rank, world, uid[0])
        if overlap:
            self.staging = [torch.empty_like(self.shard) for _ in range(2)]
            self.comm_stream = torch.cuda.Stream()
            self.ev_ready = [torch.cuda.Event() for _ in range(2)]
            self.ev_free = [torch.cuda.Event() for _ in range(2)]

    def _gather_unshard(self, src, frame):
        if self.backend == "rccl-abi":
            self.comm.exchange(src, self.gathered, frame)                # ncclAllGather + un-permute behind the C ABI
            return
        if self.backend == "nccl":
            self.dist.all_gather_into_tensor(self.gathered, src)        # RCCL over xGMI
        else:                                                            # gloo rehearsal: staged through the host
            host = self.torch.empty(self.gathered.numel(), dtype=self.torch.float32)
            self.dist.all_gather_into_tensor(host, src.cpu())
            self.gathered.copy_(host)
        self.b.unshard(self.gathered, frame, self.world)

    def exchange(self, i, shard=None, frame=None):
        """Collects every rank's shard of step i into `frame`; with overlap it runs on the comm stream."""
        torch = self.torch
        shard = self.shard if shard is None else shard
        frame = self.frame if frame is None else frame
        if not self.overlap:
            self._gather_unshard(shard, frame)
            return
        s = i & 1
        cur = torch.cuda.current_stream()
        cur.wait_event(self.ev_free[s])             # staging[s] was last read by the exchange two steps back
        self.staging[s].copy_(shard)                 # snapshot: the next render keeps adding into `shard`
        self.ev_ready[s].record(cur)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(self.ev_ready[s])
            self._gather_unshard(self.staging[s], frame)
            self.ev_free[s].record(self.comm_stream)

    def drain(self):
        if self.overlap:
            self.torch.cuda.current_stream().wait_stream(self.comm_stream)

    def reset(self):
        self.shard.zero_()
        self.shard.view(-1, 4)[:, 3] = 1.0


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_command(n, argv, port=None):
    """The command `python bench.py --gpus n ...` starts as a child when it is not already running under a launcher: one
    process per GPU on this node, rendezvous on the loopback address (the container's hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), os.path.abspath(__file__), *argv]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short timing of C2 / C4 / C5 / Cornell at N = 1")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes (quote the committed summary)")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "rccl-abi", "gloo"],
                    help="rccl-abi = the library's bt_comm_* entry points; gloo = rehearsal of the N>1 path on a box with "
                         "fewer GPUs than ranks (shards staged through the host)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--no-overlap", action="store_true", help="run the frame exchange on the render stream")
    ap.add_argument("--verify", action="store_true", help="(default for N > 1) compare one extra sharded + exchanged step with "
                    "the same step rendered by one rank alone")
    ap.add_argument("--no-verify", action="store_true", help="N > 1: skip the verification step")
    ap.add_argument("--verify-all", action="store_true", help="compare the whole accumulated frame of warmup + timed steps with a "
                    "one-rank render of the same steps (costs N x the timed work on every rank)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 code path (process group, shard, all-gather, un-permute) even with one rank")
    ap.add_argument("--print-launch", action="store_true", help="print the child command a bare `--gpus N` run would start, as JSON, and exit")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.print_launch):
        # typed from a bare shell: this process becomes the launcher.  It has not imported torch, has not touched the GPU and
        # never will -- the ranks are CHILD processes (no exec), rank 0's JSON line passes through on the inherited stdout
        cmd = launch_command(args.gpus, [a for a in sys.argv[1:] if a != "--print-launch"])
        if args.print_launch:
            print(json.dumps({"launch": cmd, "cwd": ROOT}))
            return 0
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
        return subprocess.run(cmd, cwd=ROOT, env=env).returncode

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world                                  # under a launcher the environment is authoritative
    dist_path = world > 1 or args.force_dist          # `dist_path` replaces `world > 1` below
    scene_name, w, h, base_spp = WORKLOADS[args.workload]
    spp = base_spp * world if args.scaling == "weak" else base_spp      # weak: fixed work per GPU; strong: fixed total

    # counters first: child processes under rocprofv3, while this process has not initialised the GPU yet.  N = 1: the
    # headline workload, then C2 and C4 (BASELINE configs[1] and [3]) for `other_configs`.  N > 1: rank 0 measures ITS OWN
    # shard launch (1/N of the tiles at this run's samples per step) before it joins the process group; the other ranks wait
    # for it in the rendezvous.
    pmc, pmc_other = None, {}
    if rank == 0 and not args.force_dist:
        shard = (0, world) if world > 1 else None
        if args.no_pmc:
            pmc = cached_pmc(args.workload if world == 1 else f"{args.workload}_shard{world}")
        else:
            pmc = measure_pmc(args.workload, shard=shard, spp=spp if world > 1 else None, calls=4 if world == 1 else 3)
        if world == 1 and args.workload == "C3" and not args.no_other_configs:
            for name in ("C2", "C4"):
                pmc_other[name] = cached_pmc(name) if args.no_pmc else measure_pmc(name, calls=6 if name == "C2" else 3)

    import torch
    import torch.distributed as dist

    import bendy_tracer_amd as b

    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if dist_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    census = None
    if dist_path:
        # who actually joined: one entry per rank, gathered over the process group that was just formed
        prop = torch.cuda.get_device_properties(local_rank)
        mine = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "name": prop.name,
                "pci_bus_id": getattr(prop, "pci_bus_id", None), "uuid": str(getattr(prop, "uuid", "")) or None,
                "pid": os.getpid(), "host": socket.gethostname()}
        census = [None] * dist.get_world_size()
        dist.all_gather_object(census, mine)

    scene = b.Scene.load(os.path.join(ROOT, "scenes", f"{scene_name}.json.gz"))
    cam = scene.find_by_tag("camera")
    scene.set_camera_aspect(cam, w / h)                     # main.rs:218-223
    tracer = b.Tracer.with_config(b.Config(chunks_x=8, chunks_y=4))
    rc = b.RenderConfig.with_samples(spp)
    if not dist_path:
        frame = b.Buffer.new(w, h)
    else:
        ex = ShardExchange(b, torch, dist, w, h, rank, world, args.backend, overlap=not args.no_overlap)
        frame = ex.frame

    def step(i, exchange=True):
        if not dist_path:
            tracer.render(scene, cam, rc, frame, seed=SEED, sample_base=i * spp)
        else:
            tracer.render_shard(scene, cam, rc, ex.shard, w, h, rank, world, seed=SEED, sample_base=i * spp)
            if exchange:
                ex.exchange(i)

    def sync():
        if dist_path:
            ex.drain()
        torch.cuda.synchronize()
        if dist_path:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(first, exchange_every_step=True):
        """K steps bracketed by barrier + synchronize on both sides; -> (seconds, max over ranks), per-step stream ms."""
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            ev[i][0].record()                               # same stream the render kernel is launched on
            step(first + i, exchange=exchange_every_step or i == args.steps - 1)
            ev[i][1].record()
        sync()
        elapsed = time.perf_counter() - t0
        if dist_path:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend != "gloo" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, [a.elapsed_time(c) for a, c in ev]

    for i in range(args.warmup):
        step(i)
    elapsed, step_ms = timed(args.warmup)

    verified, verify_mode = None, None
    if args.verify_all:
        # the frame every rank now holds must equal what one rank renders alone with the same seeds: RNG
        # is keyed by global pixel / sample index, so the image does not depend on the number of ranks
        ref = b.Buffer.new(w, h)
        for i in range(args.warmup + args.steps):
            tracer.render(scene, cam, rc, ref, seed=SEED, sample_base=i * spp)
        torch.cuda.synchronize()
        verified = bool(torch.equal(frame.data, ref.data))
        verify_mode = f"all {args.warmup + args.steps} steps"
        del ref
    elif dist_path and (args.verify or not args.no_verify):
        # default for N > 1: ONE extra step, outside the timed region, rendered sharded + exchanged into a fresh frame and
        # by this rank alone into another; every rank compares its own gathered copy, the verdict is the AND over the ranks
        i_v = args.warmup + args.steps
        keep = ex.shard.clone()
        ex.reset()
        got = b.Buffer.new(w, h)
        tracer.render_shard(scene, cam, rc, ex.shard, w, h, rank, world, seed=SEED, sample_base=i_v * spp)
        ex.exchange(i_v, frame=got)
        ex.drain()
        ref = b.Buffer.new(w, h)
        tracer.render(scene, cam, rc, ref, seed=SEED, sample_base=i_v * spp)
        torch.cuda.synchronize()
        ok = torch.tensor([1 if torch.equal(got.data, ref.data) else 0], dtype=torch.int32,
                          device="cuda" if args.backend != "gloo" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        verified = bool(int(ok.item()))
        verify_mode = f"one extra step (sample_base {i_v * spp}, {spp} spp), every rank's gathered copy, AND over ranks"
        ex.shard.copy_(keep)
        del got, ref, keep

    final_only = None
    if dist_path and args.scaling == "strong":
        # second timed pass: the shards keep accumulating, ONE exchange after the last step (the final framebuffer)
        e2, _ = timed(args.warmup + args.steps, exchange_every_step=False)
        final_only = {"value": round(w * h * spp * args.steps / e2 / 1e6, 2), "unit": "Msamples/s",
                      "ms_per_step": round(e2 / args.steps * 1e3, 4), "exchanges": 1}

    # segment counts and the library's own HIP-event kernel times: replay the same renders, untimed
    kernel_ms, segments = [], []
    for i in range(args.steps):
        if not dist_path:
            tracer.render(scene, cam, rc, frame, seed=SEED, sample_base=(args.warmup + i) * spp)
        else:
            tracer.render_shard(scene, cam, rc, ex.shard, w, h, rank, world, seed=SEED, sample_base=(args.warmup + i) * spp)
        st = scene.last_stats()
        kernel_ms.append(st.kernel_ms)
        segments.append(st.segments)
    last = scene.last_stats()
    my_pixels = last.pixels

    total_samples = w * h * spp * args.steps
    value = total_samples / elapsed / 1e6
    out = {
        "metric": "Msamples/s (pixels x spp / s) at 1080p" if h == 1080 else "Msamples/s (pixels x spp / s)",
        "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {scene_name}.json.gz {w}x{h}x{spp}spp Subsample::None, Config=main.rs "
                               f"(max_bounces 8, max_volume_bounces 32, clip 0.01..1000, volume_step 0.1, Output::Full), "
                               f"flat space (the reference has no lens code), seed 0x5EED",
                   "samples_per_step": w * h * spp,
                   "parallelism": (f"tiles{world}:{args.backend}" + ("" if args.no_overlap else "+overlapped-allgather")) if dist_path else "single"},
    }
    if rank == 0:
        k_ms = statistics.mean(kernel_ms)
        seg = statistics.mean(segments)
        # template arguments <OUTPUT, LENS, RECTS, VOLS, PACKED>, as rocprofv3 prints them
        kernel_name = "bt_render_kernel<0, false, %s, %s, %s>" % (
            {"scene": ("false", "false"), "volume": ("false", "true"), "cloud": ("false", "true")}.get(scene_name, ("true", "false"))
            + ("true" if last.packed else "false",))
        roof = {"bound": "valu", "unit": "wave64 VALU instructions per SIMD-cycle", "peak": VALU_PEAK, "achieved": None,
                "frac": None, "traffic": None, "kernel": kernel_name, "kernel_ms": round(k_ms, 4), "slices": last.slices,
                "workgroups": last.workgroups, "packed": bool(last.packed),
                "launches_per_step": last.launches, "scratch_bytes": last.scratch_bytes,
                "render_stream_ms_per_step_timed_region": round(statistics.mean(step_ms), 4),
                "segments_per_launch": int(seg), "segments_per_sample": round(seg / (my_pixels * spp), 4)}
        roof.update(pmc_fractions(pmc, k_ms))
        # SURVEY 8(d)'s byte model, kept as a labelled non-headline figure: a wavefront formulation would move these bytes,
        # the shipped register-resident kernel does not
        alg_bytes = BYTES_PER_SEGMENT * seg + BYTES_PER_PIXEL * my_pixels
        roof["model"] = {"bytes_per_launch": int(alg_bytes), "model_frac": round(alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "note": "SURVEY 8(d) wavefront byte model (128 B per path segment + 16 B per pixel) over kernel time / "
                                 "8 TB/s; NOT traffic of this kernel, which keeps ray state in registers -- see `traffic`"}
        roof["note"] = ("VALU-bound path tracer: frac = measured wave64 VALU instructions per SIMD-cycle / 0.5 (every instruction at "
                        "the guide's 2 cycles: the lower bound); issue_cost_guide_frac = the guide's prices (2 plain, 4 transcendental "
                        "and half-rate integer); issue_cost_weighted_frac = the PMC opcode classes x the mixed-stream issue cycles "
                        "measured by tools/valu_microbench.hip (profiles/valu_issue_costs.json: the upper end; "
                        "issue_cost_pure_stream_bound prices every opcode as a stream of its own and can exceed 1); fp32_flop_frac = "
                        "FP32 operations on active lanes / 157.3 TF/s; hbm_measured_frac = PMC bytes / kernel time / 8 TB/s"
                        + ("; N > 1: counters of rank 0's own shard launch, taken before it joined the process group" if world > 1 else ""))
        out["roofline"] = roof
        if census is not None:
            out["ranks_joined"] = dist.get_world_size()
            out["devices"] = census
            out["distinct_devices"] = len({(c["host"], c["pci_bus_id"] or c["device"]) for c in census})
        if verified is not None:
            out["verified_vs_single_rank"] = verified
            out["verify"] = verify_mode
        if final_only is not None:
            out["final_gather_only"] = final_only
        if world == 1 and args.workload == "C3" and not args.force_dist:
            out["lens_extension"] = lens_extension_rate(b, torch, scene_name, w, h)
            if not args.no_other_configs:
                out["other_configs"] = other_configs(b, torch, pmc_by_config=pmc_other)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene_name, w, h, args.cpu_budget)
            out["parity"] = parity_figure(b, torch, scene_name)
        print(json.dumps(out), flush=True)
    if dist_path:
        dist.barrier()
        if getattr(ex, "comm", None) is not None:
            ex.comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
