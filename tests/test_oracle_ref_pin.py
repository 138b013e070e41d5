"""Pin the C oracle against the reference's own sources run on the host (oracle/_ref).

Runs wherever oracle/_ref/libgsr_ref.so exists or can be built (the build container, where
/root/reference is mounted).  Everything is compared bit for bit, on seeded scenes and on
hypothesis-generated ones (random sizes, degrees, image shapes, scale modifiers, backgrounds).
"""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

from autovfx_amd import scenes
from autovfx_amd.cameras import orbit_cameras, sugar_orbit_cameras
from oracle import cpu_oracle, ref_oracle

from helpers import oracle_kwargs
from test_golden import assert_bit_identical

pytestmark = pytest.mark.skipif(not ref_oracle.available(), reason="oracle/_ref not built and /root/reference absent")


def both(kw, name):
    a = cpu_oracle.forward(intermediates=True, **kw)
    b = ref_oracle.forward(intermediates=True, **kw)
    assert_bit_identical(a, b, name)
    return a


def test_c1_full():
    both(oracle_kwargs(scenes.config_c1(), scenes.c1_camera(), bg=(0.1, 0.2, 0.3)), "c1")


def test_orbit_cloud_and_precomputed_colours():
    cam = orbit_cameras(16, 240, 135)[5]
    both(oracle_kwargs(scenes.config_c2(P=20_000, seed=3), cam), "c2-20k")
    both(oracle_kwargs(scenes.config_c4(P=8_000, seed=4), cam, bg=(1, 1, 1)), "c4-8k")


def test_c4_sugar_camera_with_off_centre_principal_point():
    """BASELINE configs[3]: the projection matrix SuGaR builds (sugar_model.py:2023-2032) through the reference's own
    kernels and the oracle, forward and backward."""
    from test_oracle_backward import assert_bits, pixel_grads
    cam = sugar_orbit_cameras(12, 200, 112, cx_ndc=0.09, cy_ndc=-0.05)[5]
    kw = oracle_kwargs(scenes.config_c4(P=12_000, seed=9), cam)
    both(kw, "c4-sugar-pp")
    kw.update(pixel_grads(cam, 11))
    assert_bits(cpu_oracle.backward(**kw), ref_oracle.backward(**kw), "c4-sugar-pp-bw")


def test_mark_visible_matches_reference():
    cam = orbit_cameras(9, 64, 64)[2]
    pts = scenes.config_c2(P=5000, seed=6).means3D
    np.testing.assert_array_equal(
        cpu_oracle.mark_visible(pts, cam.world_view_transform, cam.full_proj_transform),
        ref_oracle.mark_visible(pts, cam.world_view_transform, cam.full_proj_transform))


def test_all_culled_and_single_gaussian():
    cam = scenes.c1_camera(40, 24)
    c = scenes.config_c1(P=50, seed=1)
    c.means3D[:, 2] = -9.0
    a = both(oracle_kwargs(c, cam, bg=(0.3, 0.6, 0.9)), "culled")
    assert a["num_rendered"] == 0 and np.allclose(a["color"][2], 0.9)
    both(oracle_kwargs(scenes.config_c1(P=1, seed=2), cam), "single")


@settings(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck))
@given(P=st.integers(1, 400), W=st.integers(1, 70), H=st.integers(1, 70), deg=st.integers(0, 4),
       seed=st.integers(0, 10_000), mod=st.sampled_from([0.5, 1.0, 2.5]), big=st.booleans(),
       bg=st.tuples(*[st.floats(0, 1, width=32)] * 3))
def test_random_scenes(P, W, H, deg, seed, mod, big, bg):
    c = scenes.config_c1(P=P, seed=seed)
    if big:
        c.scales[: max(1, P // 10)] *= 20.0
    both(oracle_kwargs(c, scenes.c1_camera(W, H), bg=bg, scale_modifier=mod, sh_degree=deg), f"hyp-{seed}")
