import numpy as np
import torch
from PIL import Image


def save_image(tensor, fp, **_kw):
    t = tensor.detach().to(torch.float32)
    if t.dim() == 4:
        assert t.shape[0] == 1, "the double handles one image (no grid)"
        t = t[0]
    arr = t.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()
    Image.fromarray(np.ascontiguousarray(arr)).save(fp)
