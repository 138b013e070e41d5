"""Double of the one torchvision call the reference's frame loop makes: ``torchvision.utils.save_image(tensor, path)``
(scene_representation.py:427).  torchvision's documented conversion: ``tensor.mul(255).add_(0.5).clamp_(0, 255)`` -> uint8,
channels last, written with PIL."""
from . import utils  # noqa: F401
