"""Frame writers: dependency-free PNG against PIL's decoder, and the three per-frame files."""
import io

import numpy as np
import pytest
import torch

from autovfx_amd import frame_io


@pytest.mark.parametrize("shape", [(7, 5, 4), (64, 33, 3), (1, 1, 4)])
def test_png_roundtrip_and_pil_agrees(shape):
    img = np.random.default_rng(1).integers(0, 256, shape).astype(np.uint8)
    data = frame_io.encode_png(img)
    np.testing.assert_array_equal(frame_io.decode_png(data), img)
    PIL = pytest.importorskip("PIL.Image")
    np.testing.assert_array_equal(np.array(PIL.open(io.BytesIO(data))), img)


def test_write_frame_outputs(tmp_path):
    g = torch.Generator().manual_seed(0)
    H, W = 12, 20
    result = {"render": torch.rand(4, H, W, generator=g), "depth": torch.rand(H, W, generator=g) * 5,
              "normal": torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1)}
    paths = frame_io.write_frame_outputs(str(tmp_path), "00007", result)
    rgba = frame_io.decode_png(open(paths["images"], "rb").read())
    want = (result["render"] * 255 + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
    np.testing.assert_array_equal(rgba, want)
    np.testing.assert_array_equal(np.load(paths["depth"]), result["depth"].numpy())
    n = frame_io.decode_png(open(paths["normal"], "rb").read())
    np.testing.assert_array_equal(n, ((result["normal"] + 1) / 2 * 255).to(torch.uint8).numpy())
    assert paths["images"].endswith("images/00007.png") and paths["depth"].endswith("depth/00007.npy")
