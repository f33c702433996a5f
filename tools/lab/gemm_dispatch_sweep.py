#!/usr/bin/env python3
"""Lab: every kernel form a forward-layout bf16x3 GEMM can take (general forms 1-3, the persistent 256x256 kernel) at the
training / trimmed-sampling shapes, interleaved rounds in ONE process; form 0 = what the dispatch picks."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402
pkg = __graft_entry__.load_package()
ops, lib = pkg.ops, pkg.hip.lib()
DEV = "cuda:0"


def timed(fn, reps=20):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


shapes = [(M, N, K) for M in (2048, 4096, 8192, 16384) for N, K in ((768, 768), (1024, 768), (768, 1024), (1536, 768), (2304, 768), (768, 2304), (768, 1536))]
for M, N, K in shapes:
    a = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    fn = lambda: ops.gemm(a, w, b, mode="bf16x3")   # noqa: E731
    res = {}
    for rnd in range(5):
        for name, (p_min, form) in {"dispatch": (None, 0), "form1": ("100000", 1), "form2": ("100000", 2), "form3": ("100000", 3),
                                     "persistent": ("1", 0)}.items():
            if name == "persistent" and (M % 256 or N % 256):
                continue
            # E3D_GEMM_P_MIN is read once per process: the persistent kernel is forced through e3d_gemm_kernel_select instead
            lib.e3d_gemm_general_select(form)
            prev = lib.e3d_gemm_kernel_select(-1)
            if name in ("form1", "form2", "form3"):
                lib.e3d_gemm_kernel_select(0)          # no 256x256 forms at all
            elif name == "persistent":
                lib.e3d_gemm_kernel_select(5)          # lab value: persistent for every tile count (see gemm_split.hip)
            res.setdefault(name, []).append(timed(fn))
            lib.e3d_gemm_kernel_select(prev)
            lib.e3d_gemm_general_select(0)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    best = min((v, k) for k, v in med.items() if k != "dispatch")
    print(f"M={M:5d} N={N:4d} K={K:4d}: " + "  ".join(f"{k} {v:6.1f}" for k, v in med.items()) + f"   best {best[1]}  dispatch/best {med['dispatch'] / best[0]:.2f}",
          flush=True)
