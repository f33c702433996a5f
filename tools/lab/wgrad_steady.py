#!/usr/bin/env python3
"""Steady-state rate of the K-major x K-major (weight-gradient) GEMM layout when there are enough output tiles to fill
the chip without split-K -- what a grouped launch over the layers of a model would see -- next to the per-layer
split-K launches of today."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402
pkg = __graft_entry__.load_package()
from e3diff_amd import autograd as AG  # noqa: E402
DEV = "cuda:0"


def timeit(fn, rep=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3


for (N, K, M) in ((768, 768, 4096), (2304, 768, 4096), (768, 1024, 4096), (768 * 16, 768 * 4, 4096), (2304 * 8, 768 * 4, 4096),
                  (768 * 16, 768 * 4, 8192)):
    dz = torch.randn(M, N, device=DEV); x = torch.randn(M, K, device=DEV)
    for form in (1, 2):
        pkg.hip.lib().e3d_gemm_general_select(form)
        us = timeit(lambda: AG.gemm_general(dz, True, x, True, N, K, M, mode="bf16x3"))
        print(f"dW[{N}x{K}] over M={M} form {form}: {us:8.1f} us  {2 * N * K * M / us / 1e6:6.1f} TFLOP/s (algorithmic)", flush=True)
    pkg.hip.lib().e3d_gemm_general_select(0)
