#!/usr/bin/env python3
"""Lab: s_memtime stamps inside the row-complete GEMM + LayerNorm kernel (needs a -DROWLN_STAMPS build:
    tools/lab/build_variant.sh rowlnstamps "-DROWLN_STAMPS" gemm_rowln.hip && python tools/lab/rowln_stamps.py)"""
import ctypes, os, sys
import torch
from ctypes import c_float, c_int, c_int64, c_void_p as P
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
h = ctypes.CDLL(os.path.join(ROOT, "lab_build", os.environ.get("LAB_LIB", "libe3d_rowlnstamps.so")))
h.e3d_gemm_residual_layernorm_f32_split.restype = c_int
h.e3d_gemm_residual_layernorm_f32_split.argtypes = [P, c_int64, P, P, P, c_int64, P, P, c_float, P, c_int64, c_int, c_int, c_int, c_int, c_float, P]
h.e3d_weight_planes_f32_split.restype = c_int
h.e3d_weight_planes_f32_split.argtypes = [P, c_int, c_int, c_int, P, P]
DEV = "cuda:0"
M, K, H = 65536, int(os.environ.get("ROWLN_K", "768")), 768
a = torch.randn(M, K, device=DEV); w = (torch.randn(H, K, device=DEV) / K ** 0.5).contiguous()
b, res = torch.randn(H, device=DEV), torch.randn(M, H, device=DEV)
gamma, beta = torch.rand(H, device=DEV) + 0.5, torch.randn(H, device=DEV)
planes = torch.empty(H * K * 4, dtype=torch.uint8, device=DEV); out = torch.empty(M, H, device=DEV)
s = torch.cuda.current_stream().cuda_stream
assert h.e3d_weight_planes_f32_split(w.data_ptr(), H, K, 19, planes.data_ptr(), s) == 0
for _ in range(20):
    assert h.e3d_gemm_residual_layernorm_f32_split(a.data_ptr(), K, planes.data_ptr(), b.data_ptr(), res.data_ptr(), H, gamma.data_ptr(),
                                                   beta.data_ptr(), 1e-12, out.data_ptr(), H, M, H, K, 19, 1.0, s) == 0
torch.cuda.synchronize()
steps = (ctypes.c_longlong * (2 * 64 * 8))(); tiles = (ctypes.c_longlong * (2 * 8 * 8))()
h.e3d_debug_rowln_stamps.argtypes = [P, P]
assert h.e3d_debug_rowln_stamps(steps, tiles) == 0
wg = (ctypes.c_longlong * (512 * 4))()
h.e3d_debug_rowln_wg_times.argtypes = [P]
assert h.e3d_debug_rowln_wg_times(wg) == 0
t0 = min(wg[4 * i] for i in range(256))
starts = sorted((wg[4 * i] - t0) / 100.0 for i in range(256))
ends = sorted((wg[4 * i + 2] - t0) / 100.0 for i in range(256))
durs = sorted((wg[4 * i + 2] - wg[4 * i]) / 100.0 for i in range(256))
clk = sorted((wg[4 * i + 3] - wg[4 * i + 1]) / max(1, wg[4 * i + 2] - wg[4 * i]) * 100 for i in range(256))
print(f"workgroups (us, s_memrealtime): start min/med/max {starts[0]:.1f}/{starts[128]:.1f}/{starts[-1]:.1f}  end min/med/max {ends[0]:.1f}/{ends[128]:.1f}/{ends[-1]:.1f}  "
      f"duration min/med/max {durs[0]:.1f}/{durs[128]:.1f}/{durs[-1]:.1f}  clock MHz min/med/max {clk[0]:.0f}/{clk[128]:.0f}/{clk[-1]:.0f}")
byx = [[] for _ in range(8)]
for i in range(256):
    byx[i % 8].append((wg[4 * i + 2] - wg[4 * i]) / 100.0)
byp = [[] for _ in range(4)]
for i in range(256):
    byp[(i >> 3) & 3].append((wg[4 * i + 2] - wg[4 * i]) / 100.0)
print("duration by tile-order pattern (blockIdx >> 3) & 3 (min / median / max us): " + "  ".join(f"{min(v):.1f}/{sorted(v)[len(v) // 2]:.1f}/{max(v):.1f}" for v in byp))
print("duration by blockIdx % 8 (median us): " + " ".join(f"{sorted(v)[len(v) // 2]:.1f}" for v in byx))
# the step stamps hold the LAST tile of the workgroup that wrote them (64 rows: NM = 2): print tile-level first
names = ["tile top", "loads landed + barrier", "k loop done", "residual done", "stats done", "stores issued", "stores retired"]
for g in range(2):
    print(f"wave {4 * g}: tile phases (cycles)")
    for t in range(3):
        r = [tiles[(g * 8 + t) * 8 + i] for i in range(7)]
        print(f"  tile {t}: " + "  ".join(f"{names[i + 1]}={r[i + 1] - r[i]}" for i in range(6)) + f"  | total {r[6] - r[0]}")
for g in range(2):
    print(f"wave {4 * g}: k16 steps of the last tile: frags ready / compute issued / vmcnt passed / (barrier) -- cycles since step start; total to next step")
    for st in range(min(48, 2 * (K // 32))):
        r = [steps[(g * 64 + st) * 8 + i] for i in range(5)]
        nxt = steps[(g * 64 + st + 1) * 8] if st + 1 < 2 * (K // 32) else (r[4] if st & 1 else r[3])
        bar = f" barrier={r[4] - r[0]:5d}" if st & 1 else ""
        print(f"  step {st:2d}: frags={r[1] - r[0]:5d} issued={r[2] - r[0]:5d} vmcnt={r[3] - r[0]:5d}{bar} | total {nxt - r[0]:5d}   (wave0 - wave4 start skew {steps[(0 * 64 + st) * 8] - steps[(1 * 64 + st) * 8]:6d})")
