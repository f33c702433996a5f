#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels on one GPU (interleaved rounds in ONE process, median).

    python tools/bench_kernels.py gemm        # GEMM modes f32 / bf16x3 / bf16x6 on the model's shapes
    python tools/bench_kernels.py attn        # fused attention, rel-key and cross, L = 64/128/256
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
ops = pkg.ops
DEV = "cuda:0"


def time_ms(fn, iters=10, rounds=5):
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / iters)
    return sorted(out)[len(out) // 2]


def bench_gemm():
    shapes = [(65536, 768, 768), (65536, 2304, 768), (65536, 1536, 768), (65536, 1024, 768), (65536, 768, 1024),
              (65536, 3072, 768), (65536, 768, 3072), (4096, 2304, 768), (256, 4608, 768)]
    for M, N, K in shapes:
        a = torch.randn(M, K, device=DEV)
        w = torch.randn(N, K, device=DEV) / K ** 0.5
        b = torch.randn(N, device=DEV)
        ref = (a[:512].double() @ w.double().t() + b.double())
        row = f"M={M:6d} N={N:5d} K={K:5d}:"
        for mode in ("f32", "bf16x3", "bf16x6"):
            out = ops.gemm(a, w, b, mode=mode)
            err = ((out[:512].double() - ref).abs().max() / ref.abs().max()).item()
            ms = time_ms(lambda: ops.gemm(a, w, b, mode=mode))
            row += f"  {mode}: {ms:7.3f} ms {2.0 * M * N * K / ms / 1e9:7.1f} TF err {err:.1e}"
        print(row, flush=True)


def bench_attn():
    for B, L in ((1024, 64), (512, 128), (256, 256)):
        nh, H = 12, 768
        qkv = torch.randn(B * L, 3 * H, device=DEV)
        E = torch.randn(2 * L - 1, 64, device=DEV)
        mask = torch.ones(B, L, device=DEV)
        for name, e in (("relkey", E), ("plain", None)):
            ref = None
            for mode in ("f32", "bf16x3", "bf16x6"):
                fn = lambda: ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, nh, L, L, key_mask=mask,  # noqa: E731
                                           dist_emb=e, max_pos=L, mode=mode)
                out = fn()
                ref = out if ref is None else ref
                err = ((out - ref).abs().max() / ref.abs().max()).item()
                ms = time_ms(fn)
                fl = (6.0 if e is not None else 4.0) * L * L * H * B
                by = (4 * L * H * 4 + (2 * L - 1) * 256 + 4 * L) * B
                print(f"attn {name:6s} {mode:6s} B={B:5d} L={L:4d}: {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TF  "
                      f"{by / ms / 1e6:7.1f} GB/s algorithmic  rel diff vs f32 kernel {err:.1e}", flush=True)


def attn_pmc():
    """Few launches of the bench.py attention shape (B=256, L=256, rel-key) for a rocprofv3 --pmc pass."""
    B, L, nh, H = 256, 256, 12, 768
    qkv = torch.randn(B * L, 3 * H, device=DEV)
    E = torch.randn(2 * L - 1, 64, device=DEV)
    mask = torch.ones(B, L, device=DEV)
    for _ in range(5):
        ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, nh, L, L, key_mask=mask, dist_emb=E, max_pos=L)
    torch.cuda.synchronize()


def attn_shapes():
    """The rel-key attention kernel at the three sequence lengths of BASELINE.json (same token count) for a
    rocprofv3 --kernel-trace --stats run: L = 64 (per-wave kernel), 128 (4-wave cooperative), 256 (8-wave
    cooperative), default arithmetic, 20 launches each; prints the HIP-event average next to the algorithmic rates."""
    nh, H = 12, 768
    for B, L in ((1024, 64), (512, 128), (256, 256)):
        qkv = torch.randn(B * L, 3 * H, device=DEV)
        E = torch.randn(2 * L - 1, 64, device=DEV)
        mask = torch.ones(B, L, device=DEV)
        with torch.no_grad():
            fn = lambda: ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, nh, L, L, key_mask=mask,  # noqa: E731
                                       dist_emb=E, max_pos=L)
            ms = time_ms(fn, iters=4, rounds=5)
        fl = 6.0 * L * L * H * B
        by = (4 * L * H * 4 + (2 * L - 1) * 256 + 4 * L) * B
        print(f"attn relkey {ops.ATTN_MODE} B={B} L={L}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.1f} TF  {by / ms / 1e6:.0f} GB/s algorithmic "
              f"= {by / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)


def gemm_pmc():
    """Few launches of the dominant GEMM shape (65536 x 2304 x 768: the packed QKV projection, default mode)
    for a rocprofv3 --pmc pass; another N as the second argument."""
    M, N, K = 65536, int(sys.argv[2]) if len(sys.argv) > 2 else 2304, 768
    a = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    for _ in range(5):
        ops.gemm(a, w, b)
    torch.cuda.synchronize()


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    {"gemm": bench_gemm, "attn": bench_attn, "attn_pmc": attn_pmc, "gemm_pmc": gemm_pmc, "attn_shapes": attn_shapes}[what]()
