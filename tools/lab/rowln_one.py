#!/usr/bin/env python3
"""Lab: a few launches of the row-complete GEMM + LayerNorm op and of the pair it replaces (for rocprofv3 --pmc / --kernel-trace)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402
pkg = __graft_entry__.load_package()
ops = pkg.ops
ops.ROWLN_MIN_M = 1
DEV = "cuda:0"
M, H = 65536, 768
for K in (768, 1024):
    a = torch.randn(M, K, device=DEV); w = (torch.randn(H, K, device=DEV) / K ** 0.5).contiguous()
    b, res = torch.randn(H, device=DEV), torch.randn(M, H, device=DEV)
    gamma, beta = torch.rand(H, device=DEV) + 0.5, torch.randn(H, device=DEV)
    for _ in range(6):
        ops.linear_residual_layernorm(a, w, b, res, gamma, beta, 1e-12, mode="f16x3")
    if os.environ.get("ROWLN_PAIR", "0") == "1":
        for _ in range(6):
            ops.residual_layernorm(ops.gemm(a, w, b, mode="f16x3"), res, gamma, beta, 1e-12)
torch.cuda.synchronize()
