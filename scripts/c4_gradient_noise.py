"""One cold process: the C4 full-size backward case of tests/test_backward_gpu.py (200 k flat SuGaR-style Gaussians, 960x540) through the
library in its default mode (float atomics), compared the way round 4's test compared it -- max |hip - oracle_fp32| / max |oracle| per
gradient array, bar 2e-4 -- and against the fp64 truth.  Prints one JSON line.  scripts/gpu_c4_noise_hist.sh runs it N times on
a box and histograms dL_dscales: the reproduction of GPUTEST_r04's failure (2.55e-4 on the driver's box, 1.46e-4 on nine
builder runs).  The CPU oracles are deterministic: computed by the first process, cached under /tmp for the others."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from autovfx_amd import scenes                                   # noqa: E402
from autovfx_amd.cameras import sugar_orbit_cameras              # noqa: E402
from oracle import cpu_oracle                                    # noqa: E402
from helpers import oracle_kwargs                                # noqa: E402
from test_oracle_backward import pixel_grads                     # noqa: E402
from test_backward_gpu import KEYS_PRE, hip_backward             # noqa: E402

cloud, cam = scenes.config_c4(), sugar_orbit_cameras(50, 960, 540)[25]
pg = pixel_grads(cam, 8)
cache = "/tmp/c4_oracles.npz"
if os.path.exists(cache):
    z = np.load(cache)
    ref = {k[4:]: z[k] for k in z.files if k.startswith("ref_")}
    truth = {k[6:]: z[k] for k in z.files if k.startswith("truth_")}
else:
    kw = oracle_kwargs(cloud, cam, bg=(1.0, 1.0, 1.0))
    kw.update(pg)
    ref, truth = cpu_oracle.backward(**kw), cpu_oracle.backward_f64(**kw)
    np.savez(cache, **{"ref_" + k: v for k, v in ref.items() if k in KEYS_PRE}, **{"truth_" + k: v for k, v in truth.items() if k in KEYS_PRE})
mode = sys.argv[1] if len(sys.argv) > 1 else "atomic"
hip = hip_backward(cloud, cam, pg, bg=(1.0, 1.0, 1.0), mode=mode)
row = {"mode": mode}
for k in KEYS_PRE:
    a, r, t = (np.asarray(x, np.float64).reshape(-1) for x in (hip[k], ref[k], truth[k]))
    row[k] = {"vs_oracle": float(np.abs(a - r).max() / np.abs(r).max()), "vs_truth": float(np.abs(a - t).max() / np.abs(t).max()),
              "worst_gaussian_vs_oracle": int(np.argmax(np.abs(a - r)) // (a.size // cloud.P))}
print(json.dumps(row), flush=True)
