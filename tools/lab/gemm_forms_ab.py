#!/usr/bin/env python3
"""A/B of the general GEMM kernel's tile forms (e3d_gemm_general_select) in ONE process, interleaved rounds:
forward shapes of medium / small M and the weight-gradient layout.  Form 0 = the shape heuristic."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
ops, lib = pkg.ops, pkg.hip.lib()
from e3diff_amd import autograd as AG  # noqa: E402

DEV = "cuda:0"
forms = [int(x) for x in os.environ.get("FORMS", "0,1,2,3").split(",")]
MS = [int(x) for x in os.environ.get("MS", "64,256,1024,2048,4096,8192").split(",")]


def bench(fn):
    times = {f: [] for f in forms}
    outs = {}
    for f in forms:
        lib.e3d_gemm_general_select(f)
        outs[f] = fn().clone()
    same = all(torch.equal(outs[f], outs[forms[0]]) for f in forms)
    for rnd in range(7):
        for f in forms:
            lib.e3d_gemm_general_select(f)
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[f].append(e0.elapsed_time(e1) / 20)
    lib.e3d_gemm_general_select(0)
    return {f: sorted(t)[len(t) // 2] for f, t in times.items()}, same


for mode in ("bf16x3",):
    for M in MS:
        for N, K in ((768, 768), (2304, 768), (768, 1024), (1024, 768)):
            a = torch.randn(M, K, device=DEV)
            w = torch.randn(N, K, device=DEV) / K ** 0.5
            b = torch.randn(N, device=DEV)
            t, same = bench(lambda: ops.gemm(a, w, b, mode=mode))
            fl = 2.0 * M * N * K
            print(f"fwd {mode} M={M:5d} N={N:4d} K={K:4d}: " + "  ".join(f"form {f}: {t[f] * 1e3:6.1f} us {fl / t[f] / 1e9:6.1f} TF" for f in forms) + f"  same={same}", flush=True)
    for M, N, K in (() if os.environ.get("SKIP_WGRAD") else ((4096, 768, 768), (8192, 768, 768), (4096, 2304, 768), (4096, 1024, 768))):   # dW[N,K] = dz[M,N]^T x[M,K]
        dz = torch.randn(M, N, device=DEV)
        x = torch.randn(M, K, device=DEV)
        t, same = bench(lambda: AG.gemm_general(dz, True, x, True, N, K, M, mode=mode))
        fl = 2.0 * M * N * K
        print(f"wgrad {mode} tokens={M:5d} dW[{N},{K}]: " + "  ".join(f"form {f}: {t[f] * 1e3:6.1f} us {fl / t[f] / 1e9:6.1f} TF" for f in forms) + f"  same(bitwise, atomics)={same}", flush=True)
