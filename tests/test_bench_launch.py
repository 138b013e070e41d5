"""``python bench.py --gpus N`` as typed: with no launcher around it bench.py starts its own
``torch.distributed.run`` with one rank per GPU.  CPU tests of that path: the command it builds, and the whole chain --
self-launch -> torchrun -> RANK / WORLD_SIZE / MASTER_* in every rank -> process group -> ONE JSON line from rank 0 on the
parent's stdout -- with gloo standing in for RCCL (``BENCH_LAUNCH_PROBE=1``: the ranks count each other and render nothing)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launch_command_is_the_drivers_torchrun_form():
    import bench
    cmd, env = bench.launch_command(["--gpus", "8", "--steps", "20", "--warmup", "5"], 8, port=29617, environ={"PATH": "/usr/bin"})
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[3:9] == ["--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port", "29617"]
    assert cmd[9] == os.path.join(ROOT, "bench.py") and cmd[10:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/usr/bin"
    # a caller's own setting wins; a free port is picked when none is given
    cmd2, env2 = bench.launch_command([], 2, environ={"HSA_ENABLE_IPC_MODE_LEGACY": "1"})
    assert env2["HSA_ENABLE_IPC_MODE_LEGACY"] == "1" and 1024 < int(cmd2[cmd2.index("--master-port") + 1]) < 65536


def _run(env_extra, *argv, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout,
                          env=env, cwd=ROOT)


def test_dry_run_shows_the_child_and_runs_nothing():
    r = _run({"BENCH_LAUNCH_DRY_RUN": "1"}, "--gpus", "8", "--steps", "20", "--warmup", "5")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert "--nproc-per-node=8" in d["launch"] and d["launch"][-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert d["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


@pytest.mark.parametrize("n", [2, 3, 8])
def test_gpus_n_as_typed_reaches_n_ranks_and_prints_one_line(n):
    r = _run({"BENCH_LAUNCH_PROBE": "1", "BENCH_LAUNCH_ALLOW_NO_GPU": "1"}, "--gpus", str(n), "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d == {"probe": True, "n_ranks_seen": n, "ranks_counted": n, "n_gpus": n, "hsa_ipc_mode_legacy": "0"}


def test_without_enough_gpus_the_message_says_so():
    r = _run({}, "--gpus", "8", "--steps", "2")
    assert r.returncode != 0 and "this node exposes" in (r.stderr + r.stdout)
