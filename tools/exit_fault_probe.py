#!/usr/bin/env python3
"""Probe for the fault at process exit under rocprofv3: runs one library call next to torch and
prints the process's memory map at exit, so that the addresses of the fault's stack trace can be
attributed to libraries.  usage: exit_fault_probe.py gemm|lu|trsm"""
import atexit, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eigen_amd
from eigen_amd.device import DeviceMatrix, gemm_dev, trsm_dev
from eigen_amd import lu as ealu


def maps():
    keep = ("libeigen_amd", "libamdhip", "libhsa", "librocprof", "libtorch", "libc10", "librocm", "libc.so", "libstdc", "rocprofiler")
    seen = set()
    for line in open("/proc/self/maps"):
        f = line.split()
        if len(f) >= 6 and "x" in f[1] and any(k in f[5] for k in keep):
            print("MAP", f[0], f[5], flush=True)


atexit.register(maps)
what = sys.argv[1]
n = 2048
a = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(1)
if what == "gemm":
    c = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(2)
    gemm_dev("N", "N", 1.0, a, a, 1.0, c)
elif what == "lu":
    ealu.getrf_dev(a)
elif what == "trsm":
    b = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(2)
    trsm_dev("L", "L", "N", "U", 1.0, a, b)
torch.cuda.synchronize()
print("done", what, flush=True)
