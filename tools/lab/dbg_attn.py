import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__
pkg = __graft_entry__.load_package()
from oracle import bert as obert
lib = pkg.hip.lib()
DEV = "cuda:0"
def g(s): return torch.Generator().manual_seed(s)
def run(B, nh, L, P, relkey, scale=1.0):
    H = nh * 64
    qkv = torch.randn(B * L, 3 * H, generator=g(L)) * scale
    E = torch.randn(2 * P - 1, 64, generator=g(P + 1)) if relkey else None
    mask = torch.ones(B, L)
    split = lambda x: x.reshape(B, L, nh, 64).permute(0, 2, 1, 3).double()
    q, k, v = split(qkv[:, :H]), split(qkv[:, H:2 * H]), split(qkv[:, 2 * H:])
    s = q @ k.transpose(-1, -2)
    if E is not None:
        s = s + obert.relkey_scores_literal(q, E.double(), P)
    s = s / 8.0
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * L, H).float()
    dq = qkv.to(DEV)
    out = {}
    for mode in ("f32", "bf16x6", "bf16x3", "f16x3"):
        for tau in (8.0, 0.0):
            lib.e3d_attn_rescale_tau(tau)
            got = pkg.ops.attention(dq[:, :H], dq[:, H:2 * H], dq[:, 2 * H:], B, nh, L, L, key_mask=mask.to(DEV),
                                    dist_emb=None if E is None else E.to(DEV), max_pos=P, mode=mode)
            out[(mode, tau)] = ((got.cpu() - ref).abs().max() / ref.abs().max()).item()
    lib.e3d_attn_rescale_tau(8.0)
    print(f"B={B} nh={nh} L={L} relkey={relkey} scale={scale}: " + "  ".join(f"{m}/tau{int(t)}={e:.1e}" for (m, t), e in out.items()), flush=True)
for L in (128, 256):
    for relkey in (True, False):
        for scale in (1.0, 0.5):
            run(1, 3, L, L, relkey, scale)
