"""Generate tests/golden/*.npz from the REFERENCE's own sources compiled for the host
(oracle/_ref, see oracle/build_ref.py).  Run in the build container (needs /root/reference):

    python tests/golden/make_golden.py

Each fixture stores the *inputs* (so nothing about scene generation can drift) and every output
and intermediate the reference produced for them.  tests/test_golden.py checks the C oracle
bit-for-bit against these on CPU, and tests/test_parity_gpu.py checks the HIP path against them on
the GPU box, where /root/reference does not exist.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from autovfx_amd import scenes  # noqa: E402
from autovfx_amd.cameras import orbit_cameras, sugar_orbit_cameras  # noqa: E402
from oracle import ref_oracle  # noqa: E402
from helpers import oracle_kwargs  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    c = scenes.config_c1(P=1500, seed=101)
    yield "sh3_c1mini_96x64", oracle_kwargs(c, scenes.c1_camera(96, 64), bg=(0.1, 0.2, 0.3))
    for deg in (0, 1, 2, 4):
        c = scenes.config_c1(P=600, seed=110 + deg)
        yield f"sh{deg}_48x48", oracle_kwargs(c, scenes.c1_camera(48, 48), sh_degree=deg)
    c = scenes.config_c4(P=3000, seed=120)
    yield "precomp_flat_orbit_80x45", oracle_kwargs(c, orbit_cameras(8, 80, 45)[2], bg=(1.0, 1.0, 1.0))
    c = scenes.config_c1(P=800, seed=130)
    r, x, y, z = c.rotations.unbind(1)
    R = torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)), 1).view(-1, 3, 3)
    L = R @ torch.diag_embed(c.scales)
    S = L @ L.transpose(1, 2)
    cov = torch.stack((S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]), 1).contiguous()
    yield "cov3d_precomp_50x70", oracle_kwargs(c, scenes.c1_camera(50, 70), cov3D_precomp=cov, scale_modifier=1.0)
    c = scenes.config_c1(P=900, seed=140)
    yield "scale_mod_ragged_33x17", oracle_kwargs(c, scenes.c1_camera(33, 17), scale_modifier=1.6, bg=(0.9, 0.1, 0.5))
    c = scenes.config_c2(P=2500, seed=150)
    c.scales *= 6.0
    yield "orbit_normal_cloud_112x63", oracle_kwargs(c, orbit_cameras(20, 112, 63)[7])
    c = scenes.config_c1(P=400, seed=160)
    c.means3D[:, 2] = 0.0      # one depth plane: order decided by the tie rule
    c.scales[:30] *= 25.0      # some screen-filling splats
    yield "ties_and_big_splats_64x64", oracle_kwargs(c, scenes.c1_camera(64, 64))
    # BASELINE configs[3]: SuGaR's call shape -- flat surface-bound Gaussians, colors_precomp, and the projection matrix
    # with the off-centre principal-point terms of sugar_model.py:2029-2030
    c = scenes.config_c4(P=4000, seed=170)
    yield "c4_sugar_principal_point_120x68", oracle_kwargs(c, sugar_orbit_cameras(12, 120, 68, cx_ndc=0.11, cy_ndc=-0.07)[4],
                                                           bg=(0.0, 0.0, 0.0))


def backward_cases():
    """Small forward+backward cases through the reference's backward.cu (bw_*.npz)."""
    import numpy as np
    def grads(cam, seed):
        g = np.random.default_rng(seed)
        H, W = cam.image_height, cam.image_width
        return dict(dL_dcolor=g.standard_normal((3, H, W)).astype(np.float32),
                    dL_ddepth=(0.1 * g.standard_normal((1, H, W))).astype(np.float32),
                    dL_dalpha=g.standard_normal((1, H, W)).astype(np.float32))
    cam = scenes.c1_camera(48, 32)
    kw = oracle_kwargs(scenes.config_c1(P=500, seed=201), cam, bg=(0.1, 0.2, 0.3)); kw.update(grads(cam, 1))
    yield "bw_sh3_48x32", kw
    cam = orbit_cameras(8, 40, 30)[3]
    kw = oracle_kwargs(scenes.config_c4(P=800, seed=202), cam, bg=(1.0, 1.0, 1.0)); kw.update(grads(cam, 2))
    yield "bw_precomp_flat_40x30", kw
    cam = scenes.c1_camera(33, 17)
    c = scenes.config_c1(P=300, seed=203)
    c.scales[:20] *= 20.0
    kw = oracle_kwargs(c, cam, scale_modifier=1.3, sh_degree=1, bg=(0.5, 0.5, 0.5)); kw.update(grads(cam, 3))
    yield "bw_sh1_big_splats_33x17", kw
    cam = sugar_orbit_cameras(12, 60, 34, cx_ndc=0.11, cy_ndc=-0.07)[7]
    kw = oracle_kwargs(scenes.config_c4(P=1200, seed=204), cam, bg=(0.0, 0.0, 0.0)); kw.update(grads(cam, 4))
    yield "bw_c4_sugar_principal_point_60x34", kw


def main():
    only = sys.argv[1:]   # optional: regenerate just the named fixtures
    for name, kw in backward_cases():
        if only and name not in only:
            continue
        out = ref_oracle.backward(**kw)
        inputs = {"in_" + k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v))
                  for k, v in kw.items() if v is not None}
        outputs = {"out_" + k: np.asarray(v) for k, v in out.items()}
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **inputs, **outputs)
        print(f"{name}: P={kw['means3D'].shape[0]} D={out['num_rendered']} -> {os.path.getsize(path) / 1024:.0f} KiB")
    for name, kw in cases():
        if only and name not in only:
            continue
        out = ref_oracle.forward(intermediates=True, **kw)
        inputs = {"in_" + k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v))
                  for k, v in kw.items() if v is not None}
        outputs = {"out_" + k: np.asarray(v) for k, v in out.items()}
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **inputs, **outputs)
        print(f"{name}: P={kw['means3D'].shape[0]} D={out['num_rendered']} -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
