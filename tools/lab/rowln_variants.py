#!/usr/bin/env python3
"""Lab: time e3d_gemm_residual_layernorm_f32_split of several builds (product + lab_build/libe3d_rowln<v>.so) in one process,
interleaved rounds, median.    python tools/lab/rowln_variants.py 1 2 4 ...   (M K via ROWLN_M / ROWLN_K)"""
import ctypes, os, sys
import torch
from ctypes import c_float, c_int, c_int64, c_void_p as P
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
paths = [("product", os.path.join(ROOT, "e3-invaraint-diffusion-model_amd", "libe3d_hip.so"))]
paths += [(f"lab{v}", os.path.join(ROOT, "lab_build", f"libe3d_rowln{v}.so")) for v in sys.argv[1:]]
libs = []
for n, path in paths:
    h = ctypes.CDLL(path)
    h.e3d_gemm_residual_layernorm_f32_split.restype = c_int
    h.e3d_gemm_residual_layernorm_f32_split.argtypes = [P, c_int64, P, P, P, c_int64, P, P, c_float, P, c_int64, c_int, c_int, c_int, c_int, c_float, P]
    h.e3d_weight_planes_f32_split.restype = c_int
    h.e3d_weight_planes_f32_split.argtypes = [P, c_int, c_int, c_int, P, P]
    libs.append((n, h))
DEV = "cuda:0"
M = int(os.environ.get("ROWLN_M", "65536"))
for K in (768, 1024):
    H = 768
    a = torch.randn(M, K, device=DEV); w = (torch.randn(H, K, device=DEV) / K ** 0.5).contiguous()
    b, res = torch.randn(H, device=DEV), torch.randn(M, H, device=DEV)
    gamma, beta = torch.rand(H, device=DEV) + 0.5, torch.randn(H, device=DEV)
    planes = torch.empty(H * K * 4, dtype=torch.uint8, device=DEV)
    out = torch.empty(M, H, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    assert libs[0][1].e3d_weight_planes_f32_split(w.data_ptr(), H, K, 19, planes.data_ptr(), s) == 0

    def call(h):
        rc = h.e3d_gemm_residual_layernorm_f32_split(a.data_ptr(), K, planes.data_ptr(), b.data_ptr(), res.data_ptr(), H, gamma.data_ptr(),
                                                     beta.data_ptr(), 1e-12, out.data_ptr(), H, M, H, K, 19, 1.0, s)
        assert rc == 0, rc
    times = {n: [] for n, _ in libs}
    for rnd in range(7):
        for n, h in libs:
            call(h)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call(h)
            e1.record(); torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / 10)
    print(f"M={M} K={K}: " + "   ".join(f"{n}: {sorted(t)[len(t) // 2] * 1e3:6.1f}" for n, t in times.items()) + "  (us)", flush=True)
