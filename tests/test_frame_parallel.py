"""The N > 1 path on CPU: two ``gloo`` ranks shard a trajectory round-robin, render their frames with
a stand-in renderer (the CPU oracle -- tests may use it, the product path never does) and gather to
rank 0; the result must equal the single-process run frame for frame."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from autovfx_amd import frame_parallel as fp
from autovfx_amd import scenes
from autovfx_amd.cameras import orbit_cameras


def oracle_render(cloud, cam, bg):
    from oracle import cpu_oracle
    r = cpu_oracle.forward(means3D=cloud.means3D, opacities=cloud.opacities, bg=bg.numpy(), width=cam.image_width,
                           height=cam.image_height, viewmatrix=cam.world_view_transform,
                           projmatrix=cam.full_proj_transform, campos=cam.camera_center, tanfovx=cam.tanfovx,
                           tanfovy=cam.tanfovy, sh_degree=cloud.sh_degree, shs=cloud.shs, scales=cloud.scales,
                           rotations=cloud.rotations)
    return (torch.from_numpy(r["color"]), torch.from_numpy(r["depth"]), torch.from_numpy(r["alpha"]),
            torch.from_numpy(r["radii"]))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_frames, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cloud = scenes.config_c1(P=300, seed=2)
        cams = orbit_cameras(num_frames, 40, 24)
        res = fp.render_trajectory(cloud, cams, torch.zeros(3), keep_depth=True, render_fn=oracle_render)
        if rank == 0:
            torch.save(res, out_path)
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


def _worker_broadcast(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cloud = scenes.config_c4(P=200, seed=6) if rank == 0 else None      # only rank 0 "has the file"
        got = fp.broadcast_cloud(cloud, src=0)
        torch.save({k: getattr(got, k) for k in ("means3D", "opacities", "scales", "rotations", "shs", "colors_precomp",
                                                   "sh_degree")}, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_broadcast_cloud_replicates_rank0(tmp_path):
    mp.spawn(_worker_broadcast, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    want = scenes.config_c4(P=200, seed=6)
    for r in range(2):
        got = torch.load(str(tmp_path / f"rank{r}.pt"))
        for k in ("means3D", "opacities", "scales", "rotations", "colors_precomp"):
            assert torch.equal(got[k], getattr(want, k)), (r, k)
        assert got["shs"] is None and got["sh_degree"] == want.sh_degree
    assert fp.broadcast_cloud(want) is want      # no process group: identity


def _worker_pipelined(rank, world, port, per_rank, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cloud = scenes.config_c1(P=300, seed=2)
        cams = orbit_cameras(world * per_rank, 40, 24)
        mine = [i * world + rank for i in range(per_rank)]          # what bench.py gives rank `rank`
        res = fp.render_and_gather(cloud, cams, mine, torch.zeros(3), render_fn=oracle_render, chunks=3)
        if rank == 0:
            torch.save(res, out_path)
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("per_rank", [5, 2])
def test_pipelined_gather_two_ranks(tmp_path, per_rank):
    """render_and_gather: chunked asynchronous gathers behind the rendering give every rank's frames, in order."""
    out = str(tmp_path / "gathered.pt")
    mp.spawn(_worker_pipelined, args=(2, _free_port(), per_rank, out), nprocs=2, join=True)
    got = torch.load(out)
    cloud = scenes.config_c1(P=300, seed=2)
    cams = orbit_cameras(2 * per_rank, 40, 24)
    assert got.shape == (2, per_rank, 4, 24, 40) and got.dtype == torch.uint8
    for r in range(2):
        ref = fp.render_and_gather(cloud, cams, [i * 2 + r for i in range(per_rank)], torch.zeros(3),
                                   render_fn=oracle_render, chunks=1)
        assert ref.shape == (1, per_rank, 4, 24, 40) and torch.equal(got[r], ref[0])


class _FakePending:
    """GPU-shaped stand-in for ``rasterize_begin``'s result: the work happens in ``finish()``, and the object records
    in which order the driver began and finished frames."""
    log = []

    def __init__(self, cloud, cam, bg, tag):
        self.args, self.tag = (cloud, cam, bg), tag
        _FakePending.log.append(("begin", tag))

    def finish(self):
        _FakePending.log.append(("finish", self.tag))
        return oracle_render(*self.args)


def _fake_begin(cloud, cam, bg):
    return _FakePending(cloud, cam, bg, float(cam.camera_center[0]))


def _worker_uneven(rank, world, port, num_frames, split, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cloud = scenes.config_c1(P=300, seed=2)
        cams = orbit_cameras(num_frames, 40, 24)
        mine = fp.shard_frames(num_frames, rank, world)               # lengths differ when world does not divide F
        stats = {}
        kw = dict(begin_fn=_fake_begin, driver="pipelined", streams=3) if split else dict(render_fn=oracle_render)
        res = fp.render_and_gather(cloud, cams, mine, torch.zeros(3), chunks=3, stats=stats, **kw)
        assert stats["render_s"] >= 0.0 and stats["gather_tail_s"] >= 0.0
        if split:   # three frames in flight: the third is begun before the first is finished
            kinds = [k for k, _ in _FakePending.log]
            assert kinds[:min(3, len(mine))] == ["begin"] * min(3, len(mine))
            assert kinds.count("begin") == kinds.count("finish") == len(mine)
        if rank == 0:
            torch.save(fp.frames_in_order(res, num_frames), out_path)
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_frames,split", [(5, False), (7, True), (1, True), (6, True)])
def test_uneven_shards_and_pipelined_driver_two_ranks(tmp_path, num_frames, split):
    """render_and_gather with F not divisible by the world size (zero-padded last piece), through the blocking
    render_fn and through the pipelined driver (split calls, three in flight, on_chunk + asynchronous gathers) fed by
    a GPU-shaped fake: frame for frame what one process renders."""
    out = str(tmp_path / "gathered.pt")
    mp.spawn(_worker_uneven, args=(2, _free_port(), num_frames, split, out), nprocs=2, join=True)
    got = torch.load(out)
    cloud = scenes.config_c1(P=300, seed=2)
    cams = orbit_cameras(num_frames, 40, 24)
    ref = fp.render_trajectory(cloud, cams, torch.zeros(3), render_fn=oracle_render)["rgba8"]
    assert got.shape == (num_frames, 4, 24, 40) and torch.equal(got, ref)


def test_frames_in_order():
    g = torch.arange(2 * 3).reshape(2, 3, 1)          # rank-major: rank 0 holds frames 0, 2, 4; rank 1 holds 1, 3, (pad)
    g[0, :, 0] = torch.tensor([0, 2, 4]); g[1, :, 0] = torch.tensor([1, 3, 99])
    assert fp.frames_in_order(g, 5)[:, 0].tolist() == [0, 1, 2, 3, 4]
    with pytest.raises(ValueError):
        fp.frames_in_order(g, 7)


def test_shard_frames_round_robin():
    assert fp.shard_frames(10, 0, 4) == [0, 4, 8]
    assert fp.shard_frames(10, 3, 4) == [3, 7]
    assert fp.shard_frames(2, 3, 4) == []
    got = sorted(sum((fp.shard_frames(801, r, 8) for r in range(8)), []))
    assert got == list(range(801))
    with pytest.raises(ValueError):
        fp.shard_frames(4, 4, 4)


def test_pack_rgba8_is_save_image_rounding():
    x = torch.tensor([0.0, 0.5 / 255, 1.49 / 255, 0.999, 1.0, 1.7, -0.2]).view(1, 1, -1)
    q = fp.pack_rgba8(x.repeat(3, 1, 1), x)
    np.testing.assert_array_equal(q[0, 0].numpy(), [0, 1, 1, 255, 255, 255, 0])


@pytest.mark.parametrize("num_frames", [5, 4])
def test_two_rank_gloo_matches_single_process(tmp_path, num_frames):
    out = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(2, _free_port(), num_frames, out), nprocs=2, join=True)
    got = torch.load(out)
    cloud = scenes.config_c1(P=300, seed=2)
    cams = orbit_cameras(num_frames, 40, 24)
    ref = fp.render_trajectory(cloud, cams, torch.zeros(3), keep_depth=True, render_fn=oracle_render)
    assert got["rgba8"].shape == (num_frames, 4, 24, 40) and got["rgba8"].dtype == torch.uint8
    assert torch.equal(got["rgba8"], ref["rgba8"]) and torch.equal(got["depth"], ref["depth"])
    assert len({bytes(f.numpy().tobytes()) for f in got["rgba8"]}) == num_frames   # frames differ: order is checked
