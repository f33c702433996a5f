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
sub = torch.cat([qkv[:, 64*h:64*h+64], qkv[:, H+64*h:H+64*h+64], qkv[:, 2*H+64*h:2*H+64*h+64]], 1).contiguous()
q, k, v = sub[:, :64].double(), sub[:, 64:128].double(), sub[:, 128:].double()
s = q @ k.t() / 8
p = torch.softmax(s, -1)
ref = (p @ v).float()
d = sub.to(DEV)
for mode in ("f16x3", "bf16x3", "bf16x6"):
    got = pkg.ops.attention(d[:, :64], d[:, 64:128], d[:, 128:], 1, 1, L, L, key_mask=torch.ones(1, L, device=DEV), mode=mode).cpu()
    err = (got - ref).abs()
    r = int(err.max(dim=1).values.argmax())
    delta = (got[r] - ref[r]).double()
    # best single-key explanation: delta ~ c * (v_k - o)
    o = ref[r].double()
    best = None
    for kk in range(L):
        dv = v[kk] - o
        c = float(delta @ dv / (dv @ dv))
        res = float((delta - c * dv).norm() / delta.norm())
        if best is None or res < best[0]:
            best = (res, kk, c)
    print(mode, "standalone head: max err", float(err.max()), "worst row", r, "row err", float(err[r].max()),
          "| best single-key fit: residual", f"{best[0]:.3f}", "key", best[1], "dp", f"{best[2]:.3e}", "p_k", f"{float(p[r, best[1]]):.4f}",
          "score", f"{float(s[r, best[1]]):.3f}", flush=True)
