"""Regenerates autovfx_amd/frame_io.TURBO_LUT: the published 256-entry turbo colour map (matplotlib's copy of the float
table; OpenCV's colormap.cpp carries the same numbers) times 255, rounded to nearest even as cv::Mat::convertTo does.
    python scripts/make_turbo_lut.py      # prints the base64 literal and how far the closest entry is from a rounding boundary
"""
import base64

import numpy as np
from matplotlib import _cm_listed

data = np.array(_cm_listed._turbo_data, dtype=np.float32)
lut = np.rint(data * np.float32(255.0)).astype(np.uint8)
frac = (data.astype(np.float64) * 255) % 1
print("closest distance to a rounding boundary:", float(np.abs(frac - 0.5).min()))
b = base64.b64encode(lut.tobytes()).decode()
for i in range(0, len(b), 120):
    print('    "%s"' % b[i:i + 120])
