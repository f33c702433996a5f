#!/usr/bin/env python3
"""Lab: phase time stamps inside the 256x256 split GEMM (needs a -DE3D_STAMPS build of the library:
tools/lab_build_stamps.sh writes lab_build/libe3d_stamps.so; run with E3D_HIP_LIB pointing at it)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
ops, hip = pkg.ops, pkg.hip
M, N, K = 65536, 768, 768
a = torch.randn(M, K, device="cuda:0")
w = torch.randn(N, K, device="cuda:0") / K ** 0.5
b = torch.randn(N, device="cuda:0")
for _ in range(3):
    ops.gemm(a, w, b)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (2 * 32 * 8))()
fn = hip.lib().e3d_debug_read_stamps
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
assert fn(buf) == 0
names = ["iter start", "gloads issued", "ks0 mfma issued", "ks1 mfma issued", "gloads landed", "split+write issued", "barrier passed"]
for g in range(2):
    print(f"wave group {g}: per k-iteration cycles since the iteration's start (s_memtime ticks)")
    for kt in range(24):
        row = [buf[(g * 32 + kt) * 8 + s] for s in range(7)]
        nxt = buf[(g * 32 + kt + 1) * 8] if kt + 1 < 24 else row[6]
        print(f"  kt {kt:2d}: " + "  ".join(f"{n}={row[i] - row[0]:6d}" for i, n in enumerate(names) if i) + f"  | total {nxt - row[0]:6d}")
for g in range(2):
    t = [buf[(g * 32 + i) * 8 + 7] for i in range(6)]
    print(f"group {g} kernel-level: addressing {t[1]-t[0]}  prologue {t[2]-t[1]}  k-loop {t[3]-t[2]}  stores issued {t[4]-t[3]}  stores retired {t[5]-t[4]}  total {t[5]-t[0]}")
