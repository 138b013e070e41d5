#!/usr/bin/env python
"""What the GPU's zlib decoder (gsr_inflate_zlib_blocks) costs per literal, per match and per copied byte: 245 760-byte blocks of
different character, 64 copies of each in one launch (a single-wave workgroup per stream).  One JSON line."""
import ctypes
import json
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from autovfx_amd import _lib, exr  # noqa: E402

dev = torch.device("cuda", 0)
N = 245760
g = np.random.default_rng(0)
yy, xx = np.mgrid[0:16, 0:1920].astype(np.float32)
z = (3.0 + 2.0 * np.clip(1.5 - np.hypot(xx - 960, yy + 500) / 540, 0, 1) + 0.2 * np.sin((yy + 500) * 0.02)).astype(np.float16)
raw = b"".join(z[ln].tobytes() * 4 for ln in range(16))               # a depth pass's block: four identical half channels per line
cases = {
    "zeros (one run)": bytes(N),
    "period 4096 (long matches, far)": bytes(g.integers(0, 256, 4096, dtype=np.uint8)) * (N // 4096),
    "period 3 (long matches, overlapping)": b"abc" * (N // 3),
    "noise (literals only)": bytes(g.integers(0, 256, N, dtype=np.uint8)),
    "text-like (short matches)": bytes(g.integers(97, 105, N, dtype=np.uint8)),
    "EXR depth block": exr._predictor_and_deinterleave(raw),
}
lib = _lib.lib
out = {}
for name, data in cases.items():
    for level in (1, 6):
        s = zlib.compress(data, level)
        pad = (len(s) + 3) & ~3
        copies = 64
        packed = np.zeros(pad * copies + 4, np.uint8)
        jobs = np.zeros((copies, 4), np.uint32)
        for k in range(copies):
            packed[k * pad:k * pad + len(s)] = np.frombuffer(s, np.uint8)
            jobs[k] = (k * pad, len(s), k * len(data), len(data))
        d_packed, d_jobs = torch.from_numpy(packed).to(dev), torch.from_numpy(jobs.view(np.int32)).to(dev)
        d_out = torch.empty(copies * len(data), dtype=torch.uint8, device=dev)
        status = torch.empty(copies, dtype=torch.int32, device=dev)
        sp = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        run = lambda: lib.gsr_inflate_zlib_blocks(d_packed.data_ptr(), d_out.data_ptr(), d_jobs.data_ptr(), copies, status.data_ptr(), None, sp)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run()
        e1.record(); torch.cuda.synchronize()
        ok = bool((status == 0).all()) and bytes(d_out[:len(data)].cpu().numpy()) == data
        # symbols of the stream, counted by a host pass over zlib's own decoder is not available: estimate from the size
        out[f"{name}, level {level}"] = {"stream_bytes": len(s), "ms": round(e0.elapsed_time(e1) / 3, 3), "MB_per_s_per_stream": round(len(data) / (e0.elapsed_time(e1) / 3) / 1e3, 1), "ok": ok}
print(json.dumps(out))
