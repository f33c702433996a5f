#!/usr/bin/env python3
"""Lab: the LDS-DMA prototype of the 256x256 split GEMM (tools/lab/gemm_planes_glds4.hip) against the product kernel.
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC tools/lab/gemm_planes_glds4.hip -o lab_build/libgemm_glds4.so
    python tools/lab/gemm_glds4_ab.py"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402
pkg = __graft_entry__.load_package()
ops = pkg.ops
DEV = "cuda:0"
lab = ctypes.CDLL(os.path.join(ROOT, "lab_build", os.environ.get("LAB_LIB", "libgemm_glds4.so")))
lab_fn = getattr(lab, os.environ.get("LAB_FN", "lab_gemm_planes_glds4"))
lab_fn.restype = ctypes.c_int
lab_fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]


def to_planes(x, dt):
    hi = x.to(dt)
    lo = (x - hi.float()).to(dt)
    R, K = x.shape
    both = torch.stack([hi.view(R, K // 16, 16), lo.view(R, K // 16, 16)], dim=2)      # [R, K/16, 2, 16]
    return both.contiguous().view(torch.float32).view(R, K)


def timeit(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    return sorted(ts)[3]


for f16, dt, mode in ((1, torch.float16, "f16x3"), (0, torch.bfloat16, "bf16x3")):
    for M, N, K in ((65536, 768, 768), (65536, 2304, 768), (65536, 768, 1024), (65536, 1536, 768), (65536, 1024, 768)):
        a = torch.randn(M, K, device=DEV); w = torch.randn(N, K, device=DEV) / K ** 0.5
        ap, wp = to_planes(a, dt), to_planes(w, dt)
        out = torch.empty(M, N, device=DEV)
        s = torch.cuda.current_stream().cuda_stream

        def run_lab():
            rc = lab_fn(ap.data_ptr(), wp.data_ptr(), out.data_ptr(), M, N, K, f16, s)
            assert rc == 0, rc
        run_lab(); torch.cuda.synchronize()
        prod = ops.gemm(a, w, None, mode=mode)
        ref = (a[:512].double() @ w.double().t()).float()
        err = float((out[:512] - ref).abs().max() / ref.abs().max())
        diff = float((out - prod).abs().max() / prod.abs().max())
        again = out.clone(); run_lab(); torch.cuda.synchronize()
        t_lab, t_prod = timeit(run_lab), timeit(lambda: ops.gemm(a, w, None, mode=mode))
        fl = 2.0 * M * N * K
        print(f"{mode} M={M} N={N} K={K}: lab {t_lab:7.1f} us {fl / t_lab / 1e6:6.1f} TF | product {t_prod:7.1f} us {fl / t_prod / 1e6:6.1f} TF | "
              f"lab vs fp64 {err:.1e}, vs product {diff:.1e}, rerun identical {bool(torch.equal(again, out))}", flush=True)
