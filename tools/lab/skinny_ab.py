#!/usr/bin/env python3
"""Skinny (K cut across workgroups, second launch finishes the rows) vs tiled GEMM at small M.  Launches are captured into one HIP graph
of 40 dependent launches per variant, so the figure is kernel time + graph-edge latency, not host launch cost."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402
pkg = __graft_entry__.load_package()
ops = pkg.ops
lib = pkg.hip.lib()
DEV = "cuda:0"
REP = 40


def graph_time(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / REP * 1e3)
    return sorted(ts)[3]


for M in (64, 128, 256, 512, 1024, 2048):
    for N, K, act in ((768, 768, 0), (2304, 768, 0), (1024, 768, 1), (768, 1024, 0), (1536, 768, 0)):
        # a ring of distinct weights (as in a model: every launch streams its own W from HBM)
        ws = [torch.randn(N, K, device=DEV) / K ** 0.5 for _ in range(8)]
        a = torch.randn(M, K, device=DEV); b = torch.randn(N, device=DEV)
        out = torch.empty(M, N, device=DEV)
        it = [0]

        def fn():
            it[0] += 1
            ops.gemm(a, ws[it[0] % 8], b, act, out=out)
        res = {}
        ops.SKINNY_MAX_M = 0
        res["tiled"] = graph_time(fn)
        ops.SKINNY_MAX_M = 4096
        for wx2, mk in ((3, 96), (6, 96), (3, 192), (3, 384)):
            lib.e3d_gemm_skinny_plan_select(wx2, mk)
            res[f"sk{wx2 / 2:g}/{mk}"] = graph_time(fn)
        lib.e3d_gemm_skinny_plan_select(0, 0)
        if N in (768, 1024) and act == 0:
            gm, bt, rs = torch.ones(N, device=DEV), torch.zeros(N, device=DEV), torch.randn(M, N, device=DEV)

            def pair():
                it[0] += 1
                ops.residual_layernorm(ops.gemm(a, ws[it[0] % 8], b, out=out), rs, gm, bt, 1e-12)

            def fused():
                it[0] += 1
                ops.linear_residual_layernorm(a, ws[it[0] % 8], b, rs, gm, bt, 1e-12)
            res["gemm+LN"] = graph_time(pair)
            res["fused LN"] = graph_time(fused)
        print(f"M={M:3d} N={N:4d} K={K:4d} act={act}: " + "  ".join(f"{k} {v:6.1f} us" for k, v in res.items()), flush=True)
