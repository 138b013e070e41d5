"""Run by tests/test_parity_gpu.py::test_unfused_blend_build_equals_the_reference_kernels_bit_for_bit in a process of its own
with GSR_LIB = autovfx_amd/lib/libgsr_hip_unfused.so (the product built with -DGSR_UNFUSED_BLEND: the blend's one fused
multiply-add written as multiply, multiply, add).  Renders BASELINE scenes with that build -- full calls and inference calls --
and with the reference's own kernels compiled for gfx950 without contraction (oracle/_ref/libgsr_ref_hip.so) and demands the
SAME BITS in colour, depth, alpha and radii.  Prints one JSON line per case."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from autovfx_amd import _lib, scenes                                   # noqa: E402
from autovfx_amd.cameras import orbit_cameras, sugar_orbit_cameras     # noqa: E402
from autovfx_amd.frame_parallel import rasterize                       # noqa: E402
from oracle import ref_hip                                             # noqa: E402


def main():
    assert _lib.LIB_PATH.endswith("libgsr_hip_unfused.so"), _lib.LIB_PATH
    dev = torch.device("cuda", 0)
    cases = [("c1", scenes.config_c1(), scenes.c1_camera(), (0.1, 0.2, 0.3)),
             ("c2_f100", scenes.config_c2(), orbit_cameras(200, 960, 540)[100], (0.0, 0.0, 0.0)),
             ("c4_f49", scenes.config_c4(), sugar_orbit_cameras(50, 960, 540)[49], (0.0, 0.0, 0.0))]
    if "--c3" in sys.argv:
        cases.append(("c3_f400", scenes.config_c3(), orbit_cameras(800, 1920, 1080)[400], (0.0, 0.0, 0.0)))
    bad = 0
    for name, cloud, cam, bg in cases:
        cloud, cam = cloud.to(dev), cam.to(dev)
        bgt = torch.tensor(bg, device=dev)
        n_ref, c_ref, d_ref, a_ref, r_ref = ref_hip.forward(cloud, cam, bgt)
        with torch.no_grad():
            color, depth, alpha, radii = rasterize(cloud, cam, bgt)           # an inference call (what bench.py times)
        torch.cuda.synchronize()
        row = {"case": name, "num_rendered": int(n_ref), "pixels": int(alpha.numel())}
        for key, got, want in (("color", color, c_ref), ("depth", depth, d_ref), ("alpha", alpha, a_ref), ("radii", radii, r_ref)):
            diff = int((got.view(torch.int32) != want.view(torch.int32)).sum())
            row[key + "_words_differ"] = diff
            bad += diff
        print(json.dumps(row), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
