#!/usr/bin/env python3
"""A/B in ONE process (interleaved rounds, median): the row-complete GEMM + bias + residual + LayerNorm launch
(csrc/gemm_rowln.hip) against the pair it replaces (persistent split GEMM, then the residual-LayerNorm row kernel).
    python tools/lab/rowln_ab.py [M ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
ops = pkg.ops
DEV = "cuda:0"
Ms = [int(x) for x in sys.argv[1:]] or [65536, 32768, 16384]
ops.ROWLN_MIN_M = 1
for mode in ("f16x3", "bf16x3"):
    for M in Ms:
        for K in (768, 1024):
            H = 768
            a = torch.randn(M, K, device=DEV)
            w = (torch.randn(H, K, device=DEV) / K ** 0.5).contiguous()
            b, res = torch.randn(H, device=DEV), torch.randn(M, H, device=DEV)
            gamma, beta = torch.rand(H, device=DEV) + 0.5, torch.randn(H, device=DEV)

            def fused():
                return ops.linear_residual_layernorm(a, w, b, res, gamma, beta, 1e-12, mode=mode)

            def gemm_only():
                return ops.gemm(a, w, b, mode=mode)

            def pair():
                return ops.residual_layernorm(ops.gemm(a, w, b, mode=mode), res, gamma, beta, 1e-12)

            f, p = fused(), pair()
            diff = float((f - p).abs().max() / p.abs().max())
            arms = {"fused": fused, "pair": pair, "gemm": gemm_only}
            times = {k: [] for k in arms}
            for rnd in range(7):
                for k, fn in arms.items():
                    fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    times[k].append(e0.elapsed_time(e1) / 10 * 1e3)
            med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
            fl = 2.0 * M * H * K
            print(f"{mode} M={M} K={K}: fused {med['fused']:7.1f} us ({fl / med['fused'] / 1e6:5.1f} TF) | pair {med['pair']:7.1f} us "
                  f"(gemm alone {med['gemm']:7.1f}) | fused / pair = {med['fused'] / med['pair']:.3f} | max diff {diff:.1e}", flush=True)
