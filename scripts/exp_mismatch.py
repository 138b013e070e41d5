import ctypes, struct, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autovfx_amd import _lib
bad = torch.zeros(1, dtype=torch.int64, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def count(first, n):
    bad.zero_()
    tot = 0
    while n > 0:
        m = min(n, 1 << 30)
        _lib.lib.gsr_selftest_exp(first, m, bad.data_ptr(), st); first += m; n -= m
    torch.cuda.synchronize()
    return int(bad.item())
f2b = lambda f: struct.unpack("<I", struct.pack("<f", f))[0]
b2f = lambda b: struct.unpack("<f", struct.pack("<I", b))[0]
edges = [-0.0, -1e-30, -1e-10, -1e-5, -0.01, -1.0, -6.0, -20.0, -80.0, -87.0, -88.0, -100.0, -103.0, -103.9, -104.0]
for a, b in zip(edges[:-1], edges[1:]):
    lo, hi = f2b(a), f2b(b)
    print(f"[{b}, {a}]: {count(lo, hi - lo)} mismatches of {hi - lo}")
