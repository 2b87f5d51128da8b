#!/usr/bin/env python3
"""dgemm 8192^3 on streams that leave 0, 8, 16, 24, 32, 64 compute units alone (top bits of the mask: bit b = unit b / 8 of
XCD b % 8): what a reservation of compute units costs the product, and whether the cost follows the number of units or the
shader engine left with the fewest."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eigen_amd
from eigen_amd.device import DeviceMatrix, gemm_dev
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
n = 8192
a = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(1)
b = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(2)
c = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(3)
total = torch.cuda.get_device_properties(0).multi_processor_count
for free in (0, 8, 16, 24, 32, 40, 64):
    words = total // 32
    mask = (ctypes.c_uint32 * words)()
    for bit in range(total - free):
        mask[bit // 32] |= 1 << (bit % 32)
    raw = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(raw), ctypes.c_uint32(words), mask)
    assert rc == 0, rc
    s = torch.cuda.ExternalStream(raw.value)
    best = 1e30
    with torch.cuda.stream(s):
        gemm_dev("N", "N", 1.0, a, b, 1.0, c, stream=s)
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(4):
                gemm_dev("N", "N", 1.0, a, b, 1.0, c, stream=s)
            e1.record(s)
            s.synchronize()
            best = min(best, e0.elapsed_time(e1) / 4)
    tf = 2.0 * n ** 3 / best / 1e9
    print(f"{free:3d} units left alone ({total - free} used): {best:.2f} ms  {tf:.1f} TFLOP/s  {tf / (total - free):.4f} per unit used", flush=True)
    s.synchronize()
    hip.hipStreamDestroy(raw)
