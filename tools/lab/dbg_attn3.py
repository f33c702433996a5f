import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__
pkg = __graft_entry__.load_package()
DEV = "cuda:0"
g = lambda s: torch.Generator().manual_seed(s)
B, nh, L = 1, 3, 256
H = nh * 64
qkv = torch.randn(B * L, 3 * H, generator=g(L))
split = lambda x: x.reshape(B, L, nh, 64).permute(0, 2, 1, 3).double()
q, k, v = split(qkv[:, :H]), split(qkv[:, H:2 * H]), split(qkv[:, 2 * H:])
s = q @ k.transpose(-1, -2) / 8.0
p = torch.softmax(s, -1)
ref = (p @ v).permute(0, 2, 1, 3).reshape(B * L, H).float()
d = qkv.to(DEV)
got = pkg.ops.attention(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], B, nh, L, L, key_mask=torch.ones(B, L, device=DEV), mode="f16x3").cpu()
err = (got - ref).abs()
print("max err", float(err.max()), "ref max", float(ref.abs().max()))
idx = err.flatten().topk(8).indices
for i in idx:
    r, c = int(i) // H, int(i) % H
    h = c // 64
    print(f"row {r} col {c} (head {h}, dim {c % 64}) err {float(err[r, c]):.2e} ref {float(ref[r, c]):.4f} got {float(got[r, c]):.4f}  max p of that query {float(p[0, h, r].max()):.3f} max score {float(s[0, h, r].max()):.2f} min score {float(s[0,h,r].min()):.2f}")
print("per head", [f"{float(err[:, 64*h:64*h+64].max()):.1e}" for h in range(nh)])
print("per q-tile", [f"{float(err[32*i:32*i+32].max()):.1e}" for i in range(8)])
print("max |q|,|k|,|v|", float(qkv[:, :H].abs().max()), float(qkv[:, H:2*H].abs().max()), float(qkv[:, 2*H:].abs().max()))
