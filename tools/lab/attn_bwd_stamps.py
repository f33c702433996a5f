#!/usr/bin/env python3
"""Lab: s_memrealtime stamps inside the fused attention backward (needs a -DBWD_STAMPS build:
    tools/lab/build_variant.sh bwdstamps "-DBWD_STAMPS" attn_bwd_coop.hip && python tools/lab/attn_bwd_stamps.py [lig|rec|full])
Training shape B = 32, 12 heads, L = 128, rel-key; ``lig``: 5-30 valid rows per item (decoder self-attention), ``rec``: 20-128,
``full``: no padding.  dO rows of padded positions are zero, as in a training step."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["E3D_HIP_LIB"] = os.path.join(ROOT, "lab_build", "libe3d_bwdstamps.so")
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
from e3diff_amd import hip  # noqa: E402
from e3diff_amd.autograd import functional as F  # noqa: E402

DEV = "cuda:0"
nh, L, H, B = 12, 128, 768, 32
kind = sys.argv[1] if len(sys.argv) > 1 else "lig"
g = torch.Generator().manual_seed(0)
lens = {"lig": torch.randint(5, 31, (B,), generator=g), "rec": torch.randint(20, 129, (B,), generator=g),
        "full": torch.full((B,), L)}[kind]
mask = (torch.arange(L)[None, :] < lens[:, None]).float().to(DEV)
with pkg.ops.arithmetic("bf16x3"):
    qkv = torch.randn(B * L, 3 * H, device=DEV, requires_grad=True)
    E = torch.randn(2 * L - 1, 64, device=DEV, requires_grad=True)
    go = torch.randn(B * L, H, device=DEV) * mask.reshape(-1, 1)
    out = F.attention(qkv, None, B, nh, L, L, key_mask=mask, dist_emb=E, max_pos=L)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(6):
        if i == 3:
            ev[0].record()
        out.backward(go, retain_graph=True)
    ev[1].record()
    torch.cuda.synchronize()
print(f"{kind}: {ev[0].elapsed_time(ev[1]) / 3 * 1e3:.1f} us per backward (fused kernel + plane / dE-sum kernels + torch allocs)")
lib = hip.lib()
t = (ctypes.c_longlong * (2 * 512 * 4 * 8))()
lib.e3d_debug_bwd_stamps.argtypes = [ctypes.c_void_p]
assert lib.e3d_debug_bwd_stamps(t) == 0
n_wg = B * nh


def at(ph, wg, w, s):
    return t[((ph * 512 + wg) * 4 + w) * 8 + s]


t0 = min(at(ph, wg, 0, 0) for ph in range(2) for wg in range(n_wg))
names = ["entry->loads landed", "->images+delta+barrier", "->wave operands", "->tile loop", "->(dE tail)", "->stores retired"]
for ph in range(2):
    starts = sorted((at(ph, wg, 0, 0) - t0) / 100 for wg in range(n_wg))
    ends = sorted((max(at(ph, wg, w, 6) for w in range(4)) - t0) / 100 for wg in range(n_wg))
    durs = sorted((max(at(ph, wg, w, 6) for w in range(4)) - at(ph, wg, 0, 0)) / 100 for wg in range(n_wg))
    print(f"phase {ph}: workgroup start min/med/max {starts[0]:.1f}/{starts[n_wg // 2]:.1f}/{starts[-1]:.1f} us, end "
          f"{ends[0]:.1f}/{ends[n_wg // 2]:.1f}/{ends[-1]:.1f}, duration {durs[0]:.1f}/{durs[n_wg // 2]:.1f}/{durs[-1]:.1f}")
    for w in range(4):
        seg = []
        for s in range(6):
            d = sorted((at(ph, wg, w, s + 1) - at(ph, wg, w, s)) / 100 for wg in range(n_wg))
            seg.append(f"{names[s]} {d[n_wg // 2]:.2f} (max {d[-1]:.2f})")
        print(f"   wave {w}: " + "  ".join(seg))
