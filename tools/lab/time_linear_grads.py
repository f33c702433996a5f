import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__
pkg = __graft_entry__.load_package()
from e3diff_amd.autograd import gemm_general
ops = pkg.ops
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
for M in (4096, 8192, 16384):
    for N, K in ((768, 768), (2304, 768), (1024, 768), (768, 1024), (3072, 768)):
        dz = torch.randn(M, N, device="cuda"); W = torch.randn(N, K, device="cuda"); Wt = W.t().contiguous()
        x = torch.randn(M, K, device="cuda")
        kmaj = t(lambda: gemm_general(dz, False, W, True, M, K, N))
        fwd = t(lambda: ops.gemm(dz, Wt, None))
        wg = t(lambda: gemm_general(dz, True, x, True, N, K, M))
        f0 = t(lambda: ops.gemm(x, W, None))
        print(f"M={M} N={N} K={K}: dgrad K-major {kmaj:6.1f} us | dgrad via W^T {fwd:6.1f} us | wgrad {wg:6.1f} us | forward {f0:6.1f} us")
