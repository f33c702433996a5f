#!/usr/bin/env python3
"""Lab: the weight-gradient launch on gradients whose padded rows are zero (B = 32 items x 128 tokens; ``lig``: 5-30 valid rows per
item, ``rec``: 20-128, ``full``: none zero) -- does a k-tile of zeros cost less than a live one?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
from e3diff_amd import autograd as AG  # noqa: E402

DEV = "cuda:0"
B, L = 32, 128
M = B * L
g = torch.Generator().manual_seed(0)
tok = torch.arange(M)
for kind in ("full", "rec", "lig"):
    lens = {"lig": torch.randint(5, 31, (B,), generator=g), "rec": torch.randint(20, 129, (B,), generator=g), "full": torch.full((B,), L)}[kind]
    valid = ((tok % L) < lens[tok // L]).float()[:, None].to(DEV)
    for N, K in ((768, 768), (2304, 768), (1024, 768)):
        dz = torch.randn(M, N, device=DEV) * valid
        x = torch.randn(M, K, device=DEV)
        fn = lambda: AG.gemm_general(dz, True, x, True, N, K, M, mode="bf16x3")   # noqa: E731
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        live_tiles = float((valid.reshape(-1, 32).sum(1) > 0).float().mean())
        print(f"{kind:4s} dW[{N},{K}] over {M} tokens ({live_tiles:.2f} of the token tiles live): {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)
