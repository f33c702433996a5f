#!/usr/bin/env python3
"""A few attention-backward launches at the training shape (B=32, L=128, rel-key) for rocprofv3 --pmc passes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402
pkg = __graft_entry__.load_package()
from e3diff_amd.autograd import functional as F  # noqa: E402
DEV = "cuda:0"
nh, L, H, B = 12, 128, 768, 32
with pkg.ops.arithmetic("bf16x3"):
    qkv = torch.randn(B * L, 3 * H, device=DEV, requires_grad=True)
    E = torch.randn(2 * L - 1, 64, device=DEV, requires_grad=True)
    mask = torch.ones(B, L, device=DEV)
    go = torch.randn(B * L, H, device=DEV)
    out = F.attention(qkv, None, B, nh, L, L, key_mask=mask, dist_emb=E, max_pos=L)
    for _ in range(3):
        out.backward(go, retain_graph=True)
    torch.cuda.synchronize()
