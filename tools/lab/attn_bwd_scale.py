#!/usr/bin/env python3
"""Does the attention backward hide its latency with more waves?  Same L, growing batch: if the time grows slower than
the batch, the launches of a training step (B = 32: 1536 query-tile waves on 1024 SIMDs) are latency-chain bound."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402
pkg = __graft_entry__.load_package()
from e3diff_amd.autograd import functional as F  # noqa: E402
DEV = "cuda:0"
nh, L, H = 12, 128, 768
with pkg.ops.arithmetic("bf16x3"):
    for B in (8, 16, 32, 64, 128, 256):
        qkv = torch.randn(B * L, 3 * H, device=DEV, requires_grad=True)
        E = torch.randn(2 * L - 1, 64, device=DEV, requires_grad=True)
        mask = torch.ones(B, L, device=DEV)
        go = torch.randn(B * L, H, device=DEV)
        out = F.attention(qkv, None, B, nh, L, L, key_mask=mask, dist_emb=E, max_pos=L)
        for _ in range(2):
            out.backward(go, retain_graph=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out.backward(go, retain_graph=True)
        e1.record(); torch.cuda.synchronize()
        print(f"B={B:4d} L={L}: attention backward (A + B + C launches, + grad accumulation) {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us "
              f"= {e0.elapsed_time(e1) / 10 * 1e3 / B:6.2f} us per item", flush=True)
