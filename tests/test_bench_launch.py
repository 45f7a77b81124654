"""bench.py's launcher logic, on CPU: `python bench.py --gpus N` typed from a bare shell must start one rank per GPU as a
CHILD process (the reference's only parallelism, `chunks.into_par_iter()` tracer/mod.rs:190-197, becomes one process per
GPU here), and the counter harness must be able to profile one rank's shard launch."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def _print_launch(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv, "--print-launch"], env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 0, r.stderr.decode()
    return json.loads(r.stdout.decode().strip().splitlines()[-1])


@pytest.mark.parametrize("n", [2, 8])
def test_bare_gpus_n_builds_the_drivers_launch_command(n):
    d = _print_launch("--gpus", str(n), "--steps", "7", "--warmup", "2", "--scaling", "strong", "--workload", "C5")
    cmd = d["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and f"--nproc-per-node={n}" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"          # the container's hostname may not resolve
    port = int(cmd[cmd.index("--master-port") + 1])
    assert 1024 < port < 65536
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    # the ranks get the caller's own arguments, minus the dry-run flag
    assert cmd[script + 1:] == ["--gpus", str(n), "--steps", "7", "--warmup", "2", "--scaling", "strong", "--workload", "C5"]
    assert d["cwd"] == ROOT


def test_launcher_never_imports_torch_or_touches_the_gpu():
    """The parent must stay GPU-free (a process that has initialised the GPU must not start other programs on this pool):
    the launcher branch runs before `import torch`."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    launcher = src.index('if "WORLD_SIZE" not in os.environ')
    assert launcher < src.index("    import torch\n")
    assert "os.exec" not in src and "execv" not in src
    # and in a process where torch cannot be imported at all the dry run still works
    d = _print_launch("--gpus", "4", env={"PYTHONPATH": os.path.join(ROOT, "tests", "no_torch_stub")})
    assert "--nproc-per-node=4" in d["launch"]


def test_under_a_launcher_the_environment_wins():
    """WORLD_SIZE set (the driver's own torchrun command): no self-launch; --print-launch still only prints."""
    e = {k: v for k, v in os.environ.items()}
    e.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--help"], env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 0 and b"--print-launch" in r.stdout and b"--no-verify" in r.stdout


def test_pmc_harness_can_profile_one_ranks_shard():
    import pmc_collect
    cmd = pmc_collect.cli_command("C3", 3, "/tmp/x.json", shard=(0, 8), spp=512)
    assert cmd[cmd.index("--shard") + 1] == "0,8"
    assert cmd[cmd.index("--samples-per-call") + 1] == "512" and cmd[cmd.index("--samples") + 1] == str(3 * 512)
    assert "--shard" not in pmc_collect.cli_command("C3", 3, None, shard=(0, 1))
    # the CLI accepts the flag (usage text) -- no GPU needed for that
    r = subprocess.run([pmc_collect.CLI, "--help"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert b"--shard rank,world" in r.stderr


def test_guide_priced_fractions():
    """derive(): `valu_issue_guide_frac` prices TRANS and INT64 at 4 cycles, everything else at 2; fp32 flops count active
    lanes only (round 2's C3 counters -> 0.73 and 79 Gflop per launch, the figures of VERDICT r2)."""
    import pmc_collect
    m = {"GRBM_GUI_ACTIVE": 7113474 * 8, "SQ_INSTS_VALU": 2512883067.0, "SQ_INSTS_VALU_ADD_F32": 466843030.0,
         "SQ_INSTS_VALU_MUL_F32": 544585929.0, "SQ_INSTS_VALU_FMA_F32": 254385807.0, "SQ_INSTS_VALU_TRANS_F32": 41145420.0,
         "SQ_INSTS_VALU_INT32": 260869751.0, "SQ_INSTS_VALU_INT64": 102715127.0, "SQ_INSTS_VALU_CVT": 11223614.0,
         "SQ_THREAD_CYCLES_VALU": 0.7909 * 64 * 1e9, "SQ_ACTIVE_INST_VALU": 1e9}
    d = pmc_collect.derive(m)
    assert abs(d["valu_issue_frac"] - 0.69) < 0.005
    assert abs(d["valu_issue_guide_frac"] - 0.73) < 0.005
    assert abs(d["fp32_flops"] / 1e9 - 79.0) < 0.5
