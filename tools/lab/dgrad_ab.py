#!/usr/bin/env python3
"""Input-gradient GEMM dx = dz . W two ways: W used as it lies (B operand K-major) against W^T made contiguous first
(the forward-layout kernels + one transpose copy per weight and step)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402
pkg = __graft_entry__.load_package()
from e3diff_amd import autograd as AG  # noqa: E402
ops = pkg.ops
DEV = "cuda:0"


def timeit(fn, rep=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(rep):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / rep * 1e3)
    return sorted(ts)[2]


with ops.arithmetic("bf16x3"):
    for M in (4096, 8192):
        for N, K in ((768, 768), (2304, 768), (1024, 768), (768, 1024)):   # weight [N,K]; dz [M,N] -> dx [M,K]
            dz = torch.randn(M, N, device=DEV); w = torch.randn(N, K, device=DEV) / K ** 0.5
            wt = w.t().contiguous()
            t_tr = timeit(lambda: w.t().contiguous())
            t_fw = timeit(lambda: ops.gemm(dz, wt, None))
            t_km = timeit(lambda: AG.gemm_general(dz, False, w, True, M, K, N))
            err = float((ops.gemm(dz, wt, None) - AG.gemm_general(dz, False, w, True, M, K, N)).abs().max())
            print(f"M={M} W[{N},{K}]: transpose {t_tr:5.1f} us + forward-layout {t_fw:6.1f} us = {t_tr + t_fw:6.1f}   K-major B {t_km:6.1f} us   (max diff {err:.1e})", flush=True)
