#!/usr/bin/env python3
"""Lab: the same GEMM launches through TWO builds of the library in one process, interleaved (same box, same clocks):
    python tools/lab/gemm_two_libs_ab.py lab_build/libe3d_prev.so e3-invaraint-diffusion-model_amd/libe3d_hip.so"""
import ctypes, sys
import torch
from ctypes import c_float, c_int, c_int64, c_void_p as P
libs = []
for path in sys.argv[1:3]:
    h = ctypes.CDLL(path)
    h.e3d_gemm_bias_act_f32_split_ex.restype = c_int
    h.e3d_gemm_bias_act_f32_split_ex.argtypes = [P, c_int64, P, P, P, c_int64, c_int, c_int, c_int, c_int, c_int, P, c_float, P]
    libs.append((path.split("/")[-1], h))
DEV = "cuda:0"
for M, N, K, act in ((65536, 768, 768, 0), (65536, 2304, 768, 0), (65536, 1024, 768, 1), (65536, 768, 1024, 0)):
    a = torch.randn(M, K, device=DEV); w = torch.randn(N, K, device=DEV) / K ** 0.5; b = torch.randn(N, device=DEV)
    out = torch.empty(M, N, device=DEV)
    s = torch.cuda.current_stream().cuda_stream

    def call(h):
        rc = h.e3d_gemm_bias_act_f32_split_ex(a.data_ptr(), K, w.data_ptr(), b.data_ptr(), out.data_ptr(), N, M, N, K, act, 19, None, 1.0, s)
        assert rc == 0, rc
    times = {n: [] for n, _ in libs}
    for rnd in range(9):
        for n, h in libs:
            call(h)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call(h)
            e1.record(); torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / 10)
    print(f"M={M} N={N} K={K} act={act}: " + "   ".join(f"{n}: {sorted(t)[len(t) // 2] * 1e3:7.1f} us (min {min(t) * 1e3:.1f})" for n, t in times.items()), flush=True)
