#!/usr/bin/env python3
"""Lab: phase time stamps inside the cooperative attention kernel (needs LAB_DEFS=-DE3D_ATTN_STAMPS
tools/lab_build_stamps.sh; run with E3D_HIP_LIB=lab_build/libe3d_stamps.so)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
ops, hip = pkg.ops, pkg.hip
B, L, nh, H = 256, 256, 12, 768
qkv = torch.randn(B * L, 3 * H, device="cuda:0")
E = torch.randn(2 * L - 1, 64, device="cuda:0") if (len(sys.argv) < 2 or sys.argv[1] != "plain") else None
mask = torch.ones(B, L, device="cuda:0")
for _ in range(3):
    ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, nh, L, L, key_mask=mask, dist_emb=E, max_pos=L)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (16 * 8))()
fn = hip.lib().e3d_debug_read_attn_stamps
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
assert fn(buf) == 0
names = ["start", "stage loads issued", "S mfma issued", "T + ring skew done", "softmax done", "PV mfma issued",
         "stage split+write", "barrier passed"]
for kt in range(L // 32):
    row = [buf[kt * 8 + s] for s in range(8)]
    print(f"kt {kt}: " + "  ".join(f"{names[i]}=+{row[i] - row[i - 1]}" for i in range(1, 8)) + f"  | total {row[7] - row[0]}")
k = [buf[15 * 8 + i] for i in range(5)]
print(f"kernel-level (workgroup 1000, wave 1), cycles: prologue {k[1] - k[0]}  key sweep {k[2] - k[1]}  epilogue to stores issued "
      f"{k[3] - k[2]}  stores retired +{k[4] - k[3]}  | total {k[4] - k[0]}")
