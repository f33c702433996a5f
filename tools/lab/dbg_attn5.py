import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__
pkg = __graft_entry__.load_package()
DEV = "cuda:0"
g = lambda s: torch.Generator().manual_seed(s)
L, nh = 256, 3
H = nh * 64
qkv = torch.randn(L, 3 * H, generator=g(L))
h = 1
q, k = qkv[:, 64*h:64*h+64], qkv[:, H+64*h:H+64*h+64]
s = q.double() @ k.double().t() / 8
p = torch.softmax(s, -1)
lse_ref = torch.logsumexp(s, -1)
R = 196
for mode in ("f16x3", "bf16x3"):
    prow = torch.zeros(L, dtype=torch.double)
    for blk in range(4):
        v = torch.zeros(L, 64)
        v[64*blk:64*blk+64] = torch.eye(64)
        sub = torch.cat([q, k, v], 1).contiguous().to(DEV)
        got, lse = pkg.ops.attention(sub[:, :64], sub[:, 64:128], sub[:, 128:], 1, 1, L, L, key_mask=torch.ones(1, L, device=DEV), mode=mode, want_lse=True)
        prow[64*blk:64*blk+64] = got[R].double().cpu()
    rel = ((prow - p[R]) / p[R])
    print(mode, "row", R, "lse err", float(lse.cpu().double()[0, 0, R] - lse_ref[R]), "| p rel err: max", float(rel.abs().max()), "mean", float(rel.mean()), "std", float(rel.std()))
    top = rel.abs().topk(6).indices.tolist()
    print("   worst keys", [(kk, f"{float(rel[kk]):.2e}", f"p={float(p[R,kk]):.4f}") for kk in top])
    # compare with neighbouring row
    for R2 in (195, 197):
        prow2 = None
    # implied score error per key: ds = rel (natural log units); correlate with k rows: ds ~ dq . k / 8
    ds = rel * 8.0
    sol = torch.linalg.lstsq(k.double(), ds.unsqueeze(1)).solution.flatten()
    res = float((k.double() @ sol - ds).norm() / ds.norm())
    big = sol.abs().topk(3)
    print("   fit ds = k . dq: residual", f"{res:.3f}", "largest dq components", [(int(i), f"{float(sol[i]):.3e}", f"q={float(q[R, i]):.5f}") for i in big.indices])
