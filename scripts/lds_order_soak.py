"""Soak of gsr_selftest_lds_atomic_order: 2.1 billion wave64 returning-LDS-add instructions with every conflict density;
the radix sort's ranking (gsr_radix.hip) relies on lanes of one instruction being served in ascending lane order.
Measured on MI355X: 137 billion lane results, 0 violations."""
import ctypes, sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from autovfx_amd import _lib
bad = torch.zeros(1, dtype=torch.int64, device="cuda")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
t0 = time.time(); total = 0
for seed in range(1, 9):
    assert _lib.lib.gsr_selftest_lds_atomic_order(65536, 1024, seed * 7919, bad.data_ptr(), stream) == 0, _lib.last_error()
    total += 65536 * 4 * 1024
torch.cuda.synchronize()
print("wave-instructions", total, "lane results", total * 64, "violations", int(bad.item()), "seconds", round(time.time() - t0, 1))
