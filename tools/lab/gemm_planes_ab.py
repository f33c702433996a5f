#!/usr/bin/env python3
"""Would a "planes pipeline" pay?  The persistent 256x256 GEMM with BOTH operands handed over pre-split (hi / lo 16-bit
terms, interleaved per k-tile of 32 so that the kernel's loads are unchanged): no split arithmetic, one ds_write_b128
per item instead of two ds_write_b64.  Needs the lab build:
    tools/lab/build_variant.sh planes "-DE3D_LAB_PLANES" gemm_split.hip
    E3D_HIP_LIB=lab_build/libe3d_planes.so python tools/lab/gemm_planes_ab.py planes      (product library: no argument)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402
pkg = __graft_entry__.load_package()
ops = pkg.ops
DEV = "cuda:0"
planes = len(sys.argv) > 1 and sys.argv[1] == "planes"


def to_planes(x):
    """[R, K] fp32 -> the same bytes as [R, K] fp32: per row and k-tile of 32, 32 bf16 hi terms then 32 bf16 lo terms"""
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    R, K = x.shape
    both = torch.stack([hi.view(R, K // 32, 32), lo.view(R, K // 32, 32)], dim=2)     # [R, K/32, 2, 32] bf16
    return both.contiguous().view(torch.float32).view(R, K)


for M, N, K in ((65536, 768, 768), (65536, 2304, 768), (65536, 768, 1024), (65536, 1536, 768)):
    a = torch.randn(M, K, device=DEV); w = torch.randn(N, K, device=DEV) / K ** 0.5
    ref = (a[:512].double() @ w.double().t()).float()
    aa, ww = (to_planes(a), to_planes(w)) if planes else (a, w)
    out = ops.gemm(aa, ww, None, mode="bf16x3")
    err = float((out[:512] - ref).abs().max() / ref.abs().max())
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(aa, ww, None, mode="bf16x3")
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    t = sorted(ts)[3]
    print(f"{'planes ' if planes else 'product'} M={M} N={N} K={K}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TFLOP/s   (rel err vs fp64 {err:.1e})", flush=True)
