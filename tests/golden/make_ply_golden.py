"""Generate tests/golden/ply_ref_point_cloud_40.{ply,npz}: a point_cloud.ply WRITTEN BY THE REFERENCE's own
``GaussianModel.save_ply`` (gaussian_model.py:201-221, run through tests/ref_plyfile_stub.py) and the six parameter tensors
the reference's ``load_ply`` (:229-266) reads back from it.  Run in the build container (needs /root/reference):

    python tests/golden/make_ply_golden.py

tests/test_render_mirror.py checks ``autovfx_amd.gaussian_model`` against both files on any box.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_plyfile_stub as stub  # noqa: E402
from autovfx_amd import scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ref = stub.reference_gaussian_model()
    g = torch.Generator().manual_seed(77)
    c = scenes.config_c1(P=40, seed=77)
    m = ref.GaussianModel(3)
    # raw (pre-activation) parameters, as a trained model holds them
    m._xyz = c.means3D.clone()
    m._features_dc = c.shs[:, :1].clone()
    m._features_rest = c.shs[:, 1:].clone()
    m._scaling = torch.log(c.scales)
    m._rotation = c.rotations + 0.01 * torch.randn(40, 4, generator=g)   # not normalised on disk
    m._opacity = torch.logit(c.opacities.clamp(1e-4, 1 - 1e-4))
    path = os.path.join(HERE, "ply_ref_point_cloud_40.ply")
    m.save_ply(path)
    back = ref.GaussianModel(3)
    with stub.on_cpu():
        back.load_ply(path)
    np.savez(os.path.join(HERE, "ply_ref_point_cloud_40.npz"), **{k: getattr(back, k).detach().numpy() for k in stub.FIELDS})
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
