"""The N > 1 code path of ``bench.py`` on the one GPU a test box has: a ONE-rank RCCL process group
(``--force-distributed``: ``init_process_group("nccl")``, barriers, the max-over-ranks all-reduce, pipelined asynchronous
``dist.gather`` of the frame stack in pieces; with ``--job-frames`` also ``broadcast_cloud`` and round-robin shards) must
start on MI355X and hand rank 0 the very bytes the N = 1 path renders.  The real N = 2 behaviour (uneven shards, padding
rows, frame order) is covered on CPU with gloo in tests/test_frame_parallel.py; what this adds is that RCCL itself
initialises and gathers here, so that the first 8-GPU run does not die of an init bug.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--workload", "c2", "--warmup", "2", "--regions", "1", "--streams", "3", "--no-cpu-baseline", "--no-also",
          "--no-reference-hip", "--frames-digest"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(*extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *COMMON, *extra], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, f"bench.py {' '.join(extra)} failed ({r.returncode}):\n{r.stderr[-3000:]}"
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["frames_digest"]["nonzero_bytes"] > 0, "the frames are empty"
    return line


def test_one_rank_rccl_group_gathers_the_bytes_of_the_single_process_path():
    plain = _bench("--steps", "8")
    dist = _bench("--steps", "8", "--force-distributed", "--gather-chunks", "3")
    assert dist["config"]["gather"]["gathered_shape"] == [1, 8, 4, 540, 960]
    assert "RCCL gather" in dist["config"]["boundary"]
    assert dist["frames_digest"] == plain["frames_digest"]
    # the multi-GPU scalars ride in `config` (where the driver's record keeps scalars): every rank's rate, the gather's tail, and
    # the job against N x rank 0 alone in the same run
    c = dist["config"]
    assert c["n_ranks_seen"] == 1 and c["per_rank_frames_per_s_min"] == c["per_rank_frames_per_s_max"] == dist["per_rank_frames_per_s"][0] > 0
    assert c["gather_tail_ms"] >= 0.0 and c["single_gpu_frames_per_s_same_run"] > 0
    assert 0.3 < c["frac_of_n_x_single_gpu"] < 1.5      # one rank: the gather's cost against the same frames without it
    assert plain["config"]["frac_of_n_x_single_gpu"] is None and plain["config"]["gather_tail_ms"] is None


def test_one_rank_rccl_job_with_cloud_broadcast_and_uneven_pieces():
    plain = _bench("--job-frames", "33")
    dist = _bench("--job-frames", "33", "--force-distributed")    # 33 frames in 16 pieces: pieces of 2 and 3 frames
    assert dist["scaling"] == "strong" and dist["config"]["job_frames"] == 33
    assert dist["config"]["gather"]["gathered_shape"] == [1, 33, 4, 540, 960]
    assert dist["frames_digest"] == plain["frames_digest"]
    assert dist["frames_digest"]["shape"] == [33, 4, 540, 960]


def test_gpus_2_as_typed_without_a_launcher():
    """``python bench.py --gpus 2`` with nothing around it: bench.py starts its own two ranks (one per GPU) over RCCL and rank 0
    prints the line.  Needs two GPUs; on a one-GPU box the same chain is covered with gloo by tests/test_bench_launch.py."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU here: the self-launch chain is covered on CPU (tests/test_bench_launch.py)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", *COMMON], capture_output=True,
                       text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and len(line["per_rank_frames_per_s"]) == 2
    assert line["config"]["gather"]["gathered_shape"] == [2, 8, 4, 540, 960]
