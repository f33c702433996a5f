"""Lab: why is the 4-wave cooperative attention workgroup (E3D_ATTN_W=4) faster than the 8-wave one standalone at L = 256
(355 vs 371 us) and twice as slow inside the model step?  Standalone timings under the step's conditions, one at a time."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__
pkg = __graft_entry__.load_package()
ops, lib = pkg.ops, pkg.hip.lib()
from bench_kernels import time_ms
from helpers import synthetic_pockets
DEV = "cuda:0"
B, L, nh, H = 256, 256, 12, 768
qkv = torch.randn(B * L, 3 * H, device=DEV)
E = torch.randn(2 * L - 1, 64, device=DEV)
ones = torch.ones(B, L, device=DEV)
pk = synthetic_pockets(B, L, seed=1000)
real = pk["receptor_attn_mask"].to(DEV).contiguous()
lig = pk["ligand_attn_mask"].to(DEV).contiguous()
bound = ops.absmax(qkv)
lib.e3d_attn_skip_padded_tiles(int(os.environ.get("SKIP", "0")))
with torch.no_grad():
    for name, mask, bounds in (("ones", ones, None), ("receptor mask", real, None), ("ligand mask", lig, None), ("receptor mask + bounds", real, (bound, bound))):
        fn = lambda: ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, nh, L, L, key_mask=mask, dist_emb=E, max_pos=L, bounds=bounds)  # noqa: E731
        print(f"W={os.environ.get('E3D_ATTN_W', '8')} skip={os.environ.get('SKIP', '0')} {name}: {time_ms(fn, iters=4, rounds=5) * 1e3:.1f} us", flush=True)
    # interleaved with a large GEMM, as in the step
    a = torch.randn(B * L, 768, device=DEV); w = torch.randn(768, 768, device=DEV) / 27.7; bb = torch.randn(768, device=DEV)
    def both():
        ops.gemm(a, w, bb)
        ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, nh, L, L, key_mask=real, dist_emb=E, max_pos=L)
    g = time_ms(lambda: ops.gemm(a, w, bb), iters=4, rounds=5)
    print(f"   gemm + attention: {time_ms(both, iters=4, rounds=5) * 1e3:.1f} us (gemm alone {g * 1e3:.1f})", flush=True)
