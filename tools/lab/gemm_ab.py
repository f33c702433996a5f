#!/usr/bin/env python3
"""A/B of the large-M GEMM kernel forms in ONE process (interleaved rounds, median; cdna guide rule 24):
    python tools/lab/gemm_ab.py [forms...]      e.g. 4 3 (see e3d_gemm_kernel_select)
Checks every form bit-for-bit against form 4 first (same products, same order)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
ops, lib = pkg.ops, pkg.hip.lib()
MODE = os.environ.get("GEMM_AB_MODE", "bf16x3")
forms = [int(x) for x in sys.argv[1:]] or [4, 3]
DEV = "cuda:0"
shapes = [(65536, 768, 768, 0), (65536, 2304, 768, 0), (65536, 1024, 768, 1), (65536, 768, 1024, 0), (65536, 1536, 768, 0),
          (16384, 768, 768, 0)]
for M, N, K, act in shapes:
    a = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    outs = {}
    for f in forms:
        lib.e3d_gemm_kernel_select(f)
        outs[f] = ops.gemm(a, w, b, act, mode=MODE).clone()
    ref = outs[forms[0]]
    same = {f: bool(torch.equal(outs[f], ref)) for f in forms}
    ref64 = (a[:256].double() @ w.double().t() + b.double())
    if act == 1:
        ref64 = torch.nn.functional.gelu(ref64)
    err = {f: ((outs[f][:256].double() - ref64).abs().max() / ref64.abs().max()).item() for f in forms}
    times = {f: [] for f in forms}
    for rnd in range(7):
        for f in forms:
            lib.e3d_gemm_kernel_select(f)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.gemm(a, w, b, act, mode=MODE)
            e0.record()
            for _ in range(10):
                ops.gemm(a, w, b, act, mode=MODE)
            e1.record()
            torch.cuda.synchronize()
            times[f].append(e0.elapsed_time(e1) / 10)
    row = f"M={M} N={N} K={K} act={act}:"
    for f in forms:
        ms = sorted(times[f])[len(times[f]) // 2]
        row += f"  form {f}: {ms * 1e3:7.1f} us {2.0 * M * N * K / ms / 1e9:6.1f} TF (min {min(times[f]) * 1e3:.1f}) same={same[f]} err={err[f]:.1e} |"
    print(row, flush=True)
lib.e3d_gemm_kernel_select(4)
