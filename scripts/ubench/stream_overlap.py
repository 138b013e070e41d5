#!/usr/bin/env python
"""How far do single-workgroup kernels on different HIP streams overlap?  The batch-of-7 ``png_unfilter_kernel`` (7 workgroups, 2.4 ms)
queued K times on each of n streams: wall time against n = 1.  Prints one JSON line."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from autovfx_amd import _lib, layer_io  # noqa: E402

dev = torch.device("cuda", 0)
W, H, C = 1920, 1080, 4
raw = np.random.default_rng(0).integers(0, 256, H * (1 + W * C)).astype(np.uint8)
raw[0::1 + W * C] = np.arange(H) % 5
staged = torch.from_numpy(raw).to(dev)
lib = _lib.lib
res = {}
for n in (1, 2, 4, 8, 16):
    streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
    tables, keep = [], []
    for _s in streams:
        outs = [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(7)]
        scr = [torch.empty(lib.gsr_png_unfilter_scratch(W, H), dtype=torch.uint8, device=dev) for _ in range(7)]
        keep += outs + scr
        tables.append((_lib.PngUnfilterJob * 7)(*[_lib.PngUnfilterJob(staged.data_ptr(), W, H, C, o.data_ptr(), s.data_ptr()) for o, s in zip(outs, scr)]))
    K = 8
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            for s, t in zip(streams, tables):
                lib.gsr_png_unfilter_batch(7, ctypes.byref(t), ctypes.c_void_p(s.cuda_stream))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    res[n] = {"ms_per_batch": round(dt / (K * n) * 1e3, 3), "overlap": None}
    del keep
base = res[1]["ms_per_batch"]
for n in res:
    res[n]["overlap"] = round(base / res[n]["ms_per_batch"], 2)
print(json.dumps({"kernel": "png_unfilter_kernel, 7 workgroups, 1920x1080", "by_streams": res, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}))
