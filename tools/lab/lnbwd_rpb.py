#!/usr/bin/env python3
"""Lab: LayerNorm backward at the training shapes, rows per block (E3D_LNBWD_RPB, read once per process)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
from e3diff_amd.autograd import functional as F  # noqa: E402

DEV = "cuda:0"
for M in (4096, 8192, 1024):
    H = 768
    x = torch.randn(M, H, device=DEV, requires_grad=True)
    r = torch.randn(M, H, device=DEV, requires_grad=True)
    ga = torch.ones(H, device=DEV, requires_grad=True)
    be = torch.zeros(H, device=DEV, requires_grad=True)
    go = torch.randn(M, H, device=DEV)
    y = F.residual_layernorm(x, r, ga, be, 1e-12)
    for _ in range(3):
        y.backward(go, retain_graph=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        y.backward(go, retain_graph=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"rpb={os.environ.get('E3D_LNBWD_RPB', 'default')} M={M}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per backward call (kernel + zero-fill + torch accumulation)", flush=True)
