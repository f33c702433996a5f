#!/usr/bin/env python3
"""A few large K-major x K-major (weight-gradient layout) launches for rocprofv3 --pmc passes (tools/lab/pmc_passes.sh)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402
pkg = __graft_entry__.load_package()
from e3diff_amd import autograd as AG  # noqa: E402
DEV = "cuda:0"
N, K, M = 768 * 16, 768 * 4, 4096
dz = torch.randn(M, N, device=DEV); x = torch.randn(M, K, device=DEV)
pkg.hip.lib().e3d_gemm_general_select(1)
for _ in range(3):
    AG.gemm_general(dz, True, x, True, N, K, M, mode="bf16x3")
torch.cuda.synchronize()
