import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__
pkg = __graft_entry__.load_package()
DEV = "cuda:0"
g = lambda s: torch.Generator().manual_seed(s)
B, nh, L = 1, 1, 256
H = 64
for case in ("random", "uniform-scores", "v-const", "tile-dependence"):
    qkv = torch.randn(L, 3 * H, generator=g(1))
    if case == "uniform-scores":
        qkv[:, :H] = 0
    if case == "v-const":
        qkv[:, 2 * H:] = 1.0
    q, k, v = qkv[:, :H].double(), qkv[:, H:2 * H].double(), qkv[:, 2 * H:].double()
    s = q @ k.t() / 8
    ref = (torch.softmax(s, -1) @ v).float()
    d = qkv.to(DEV)
    for mode in ("bf16x3", "f16x3"):
        got = pkg.ops.attention(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], B, nh, L, L, key_mask=torch.ones(1, L, device=DEV), mode=mode).cpu()
        err = (got - ref).abs()
        print(case, mode, "max err", float(err.max()), "ref max", float(ref.abs().max()),
              "per q-tile", [f"{float(err[32*i:32*i+32].max()):.1e}" for i in range(8)],
              "per 16-dim", [f"{float(err[:, 16*i:16*i+16].max()):.1e}" for i in range(4)], flush=True)
# key-tile dependence: V nonzero only in one key tile
for kt in range(8):
    qkv = torch.randn(L, 3 * H, generator=g(2))
    qkv[:, :H] = 0
    vv = torch.zeros(L, H); vv[32 * kt:32 * kt + 32] = torch.randn(32, H, generator=g(3))
    qkv[:, 2 * H:] = vv
    ref = (torch.softmax(torch.zeros(L, L, dtype=torch.double), -1) @ vv.double()).float()
    d = qkv.to(DEV)
    got = pkg.ops.attention(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], B, nh, L, L, key_mask=torch.ones(1, L, device=DEV), mode="f16x3").cpu()
    print("V only in key tile", kt, "rel err", float((got - ref).abs().max() / ref.abs().max()), flush=True)
