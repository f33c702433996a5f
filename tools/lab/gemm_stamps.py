#!/usr/bin/env python3
"""Lab: s_memtime stamps inside the general GEMM kernel's k-step (needs a -DGEMM_STAMPS=<workgroup> build:
    tools/lab/build_variant.sh gemmstamps "-DGEMM_STAMPS=77" gemm_split.hip && python tools/lab/gemm_stamps.py [M N K])
Waves 0 and 4 (both on SIMD 0) of workgroup 77; per k-step: fragments landed / MFMAs + staging issued / own loads and LDS
writes retired / barrier passed -- cycles since the top of the step."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["E3D_HIP_LIB"] = os.path.join(ROOT, "lab_build", "libe3d_gemmstamps.so")
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
ops, lib = pkg.ops, pkg.hip.lib()
DEV = "cuda:0"
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 768, 768)
ops.SKINNY_MAX_M = 0
a = torch.randn(M, K, device=DEV)
w = torch.randn(N, K, device=DEV) / K ** 0.5
b = torch.randn(N, device=DEV)
for _ in range(5):
    ops.gemm(a, w, b, mode="bf16x3")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.gemm(a, w, b, mode="bf16x3")
e1.record()
torch.cuda.synchronize()
print(f"M={M} N={N} K={K}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (stamped build)")
t = (ctypes.c_longlong * (2 * 64 * 8))()
lib.e3d_debug_gemm_stamps.argtypes = [ctypes.c_void_p]
assert lib.e3d_debug_gemm_stamps(t) == 0
nk = K // 32
for g in range(2):
    print(f"wave {4 * g}:")
    for kt in range(nk):
        r = [t[(g * 64 + kt) * 8 + i] for i in range(5)]
        nxt = t[(g * 64 + kt + 1) * 8] if kt + 1 < nk else r[4]
        print(f"  k-step {kt:2d}: frags={r[1] - r[0]:5d} issued={r[2] - r[0]:5d} own work retired={r[3] - r[0]:5d} barrier passed={r[4] - r[0]:5d}"
              f" | to next step {nxt - r[0]:5d}   (start skew wave0 - wave4 {t[kt * 8] - t[(64 + kt) * 8]:6d})")
tot = t[(nk - 1) * 8 + 4] - t[0]
print(f"wave 0: {nk} k-steps in {tot} cycles of s_memtime = {tot / nk:.0f} per k-step")
