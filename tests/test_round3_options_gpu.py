"""Round-3 options that must not change a bit.  ``GSR_OPT_DEPTH_DROP``: the depth sort drops the Gaussians that emit nothing in
its first pass (gsr_radix.hip) -- compared with the plain form on the same inputs, with fresh allocations poisoned so that a
later pass that read the undefined tail of the order would show.  ``GSR_OPT_BLEND_ORDER``: which workgroup blends which tile
(gsr_internal.h BlendOrder) -- placement only."""
import numpy as np
import pytest
from autovfx_amd import scenes
from autovfx_amd.cameras import orbit_cameras

pytestmark = pytest.mark.gpu


def _scene(name):
    if name == "c1":
        return scenes.config_c1(), scenes.c1_camera()
    if name == "ragged":
        return scenes.config_c1(P=1777, seed=23), scenes.c1_camera(250, 130)
    if name == "heavy15k":
        return scenes.config_heavy(P=15_000), orbit_cameras(200, 960, 540)[3]
    if name == "c2":
        return scenes.config_c2(), orbit_cameras(200, 960, 540)[100]
    if name == "odd858":     # 33 x 26 = 858 tiles: strips of 27 tiles that end in the middle of tile rows, the last one short
        return scenes.config_c1(P=30_000, seed=5), scenes.c1_camera(517, 403)
    if name == "c2_behind":   # a camera inside the cloud: half of the Gaussians are culled, and leave the depth sort at once
        cams = orbit_cameras(200, 960, 540, radius=0.5)
        return scenes.config_c2(), cams[17]
    raise KeyError(name)


@pytest.fixture(autouse=True)
def _defaults():
    from autovfx_amd import _lib
    from diff_gaussian_rasterization import _C
    yield
    _lib.set_option(_lib.OPT_DEPTH_DROP, 1)
    _lib.set_option(_lib.OPT_BLEND_ORDER, 1)
    _C.set_alloc_poison(None)


@pytest.mark.parametrize("scene", ["c1", "ragged", "heavy15k", "c2", "c2_behind"])
def test_depth_drop_same_lists(scene):
    from autovfx_amd import _lib
    from diff_gaussian_rasterization import _C
    from helpers import hip_forward_inference, hip_forward_raw
    cloud, cam = _scene(scene)
    outs = {}
    for drop in (0, 1):
        _lib.set_option(_lib.OPT_DEPTH_DROP, drop)
        _C.set_alloc_poison("random")
        full = hip_forward_raw(cloud, cam, cull=True, bg=(0.1, 0.2, 0.3))
        inf = hip_forward_inference(cloud, cam, slabs=0, slab_first=8 if scene in ("c1", "ragged") else 40)
        _C.set_alloc_poison(None)
        outs[drop] = (full, inf)
    a, b = outs[0][0], outs[1][0]
    V = a["sorted_count"]
    assert V == b["sorted_count"] and 0 < V <= cloud.P
    if scene == "c2_behind":
        assert V < 0.7 * cloud.P, "this camera is meant to cull a large part of the cloud"
    # without the drop the culled Gaussians follow the visible ones in index order; with it that tail is undefined
    tail = np.flatnonzero((a["tight_rect"][:, 2] | a["tight_rect"][:, 3]) == 0)
    raw0 = a["depth_order"].copy()
    assert V + tail.size == cloud.P
    for k in ("radii", "point_offsets", "tiles_touched", "point_list", "tile_keys", "ranges", "n_contrib", "live_mask", "tight_rect"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=f"{scene}: {k}")
    np.testing.assert_array_equal(raw0[:V], b["depth_order"][:V], err_msg=f"{scene}: depth order of the visible Gaussians")
    for k in ("color", "depth", "alpha"):
        np.testing.assert_array_equal(a[k].view(np.uint32), b[k].view(np.uint32), err_msg=f"{scene}: {k}")
        np.testing.assert_array_equal(outs[0][1][k].view(np.uint32), outs[1][1][k].view(np.uint32), err_msg=f"{scene} inference: {k}")
    assert outs[0][1]["slab_pairs"] == outs[1][1]["slab_pairs"]


@pytest.mark.parametrize("scene", ["odd858", "heavy15k", "c2", "c2_behind"])
def test_blend_order_is_placement_only(scene):
    """GSR_OPT_BLEND_ORDER (four strips of tiles per XCD, longest tile lists first; images of more than 256 tiles) decides which
    workgroup blends which tile, nothing else: every image, ``n_contrib`` and the lists of a full call and the images of an
    inference call cut into slabs are the same bits with it on and off."""
    from autovfx_amd import _lib
    from diff_gaussian_rasterization import _C
    from helpers import hip_forward_inference, hip_forward_raw
    cloud, cam = _scene(scene)
    outs = {}
    try:
        for order in (0, 1):
            _lib.set_option(_lib.OPT_BLEND_ORDER, order)
            assert _lib.get_option(_lib.OPT_BLEND_ORDER) == order
            _C.set_alloc_poison("random")
            full = hip_forward_raw(cloud, cam, cull=True, bg=(0.3, 0.2, 0.1))
            inf = hip_forward_inference(cloud, cam, slabs=0, slab_first=8 if scene == "odd858" else 40, bg=(0.3, 0.2, 0.1))
            _C.set_alloc_poison(None)
            outs[order] = (full, inf)
    finally:
        _lib.set_option(_lib.OPT_BLEND_ORDER, 1)
    for k in ("radii", "point_list", "tile_keys", "ranges", "n_contrib"):
        np.testing.assert_array_equal(outs[0][0][k], outs[1][0][k], err_msg=f"{scene}: {k}")
    for k in ("color", "depth", "alpha"):
        np.testing.assert_array_equal(outs[0][0][k].view(np.uint32), outs[1][0][k].view(np.uint32), err_msg=f"{scene}: {k}")
        np.testing.assert_array_equal(outs[0][1][k].view(np.uint32), outs[1][1][k].view(np.uint32), err_msg=f"{scene} inference: {k}")
        np.testing.assert_array_equal(outs[1][1][k].view(np.uint32), outs[1][0][k].view(np.uint32), err_msg=f"{scene} inference vs full: {k}")
    assert outs[0][1]["slab_pairs"] == outs[1][1]["slab_pairs"]
