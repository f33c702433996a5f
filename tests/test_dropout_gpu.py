"""GPU: dropout of the training path (reference: nn.Dropout(p=0.1) in BertEmbeddings, the SELayer MLP,
BertSelfOutput/BertOutput and on the attention probabilities -- structure_model/model.py:45-47,109-117,
transformers 4.38.2).  torch's Philox stream cannot be reproduced, so parity is checked as:
  * the decisions are a Bernoulli(1-p') field with the documented p' = round(65536 p)/65536, deterministic in
    (seed, index), and identical in forward and backward;
  * with the multipliers exported by the library (e3d_attn_dropout_mask / dropout of ones), the kernels
    equal a plain fp64 torch statement of dropout-after-softmax, forward and backward, to the usual bounds.
"""
import math

import pytest
import torch

from oracle import bert as obert
from helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def g(seed):
    return torch.Generator().manual_seed(seed)


def leaf(t, dev=None, dtype=None):
    t = t.detach().clone()
    if dtype is not None:
        t = t.to(dtype)
    if dev is not None:
        t = t.to(dev)
    return t.requires_grad_(True)


@pytest.mark.parametrize("n", [1, 3, 4, 1023, 768 * 4096 + 2])
@pytest.mark.parametrize("p", [0.1, 0.5])
def test_elementwise_dropout_values_rate_and_determinism(pkg, hip, n, p):
    ops = pkg.ops
    x = torch.randn(n, device=DEV)
    thr = round(p * 65536)
    scale = 65536.0 / (65536 - thr)
    y = ops.dropout(x, p, 1234)
    kept = y != 0
    assert torch.equal(y[kept], (x * scale)[kept])           # kept values: exactly x * scale (one fp32 multiply)
    assert torch.equal(ops.dropout(x, p, 1234), y)           # same (seed, index) -> same decisions
    if n > 1000:
        assert not torch.equal(ops.dropout(x, p, 1235) != 0, kept)
        rate = 1.0 - kept.float().mean().item()
        sigma = math.sqrt(p * (1 - p) / n)
        assert abs(rate - thr / 65536) < 5 * sigma + 1e-6, (rate, thr / 65536)
    if n > 100000:
        # no visible structure along the 4-element groups the generator works in: per-field keep rates agree
        m = kept[: n // 4 * 4].view(-1, 4).float().mean(0)
        assert (m - (1 - thr / 65536)).abs().max() < 6 * math.sqrt(p * (1 - p) / (n // 4))
        # and neighbouring decisions are uncorrelated
        a, b = kept[:-1].float(), kept[1:].float()
        corr = ((a - a.mean()) * (b - b.mean())).mean() / (a.std() * b.std())
        assert abs(corr.item()) < 5 / math.sqrt(n)


def test_dropout_backward_uses_the_same_decisions(pkg, hip):
    from e3diff_amd.autograd import functional as F
    x = leaf(torch.randn(300, 768, generator=g(1)), DEV)
    torch.manual_seed(7)
    y = F.dropout(x, 0.1)
    go = torch.randn(300, 768, generator=g(2)).to(DEV)
    y.backward(go)
    assert torch.allclose(x.grad, go * torch.where(y.detach() != 0, torch.full_like(go, 65536.0 / (65536 - 6554)), torch.zeros_like(go)))
    assert F.dropout(x, 0.1, training=False) is x and F.dropout(x, 0.0) is x


def ref_attention_dropout(q, k, v, mask, E, P, mult):
    s = q @ k.transpose(-1, -2)
    if E is not None:
        s = s + obert.relkey_scores_literal(q, E, P)
    s = s / 8.0 + ((1.0 - mask) * -10000.0)[:, None, None, :]
    return (torch.softmax(s, -1) * mult) @ v


@pytest.mark.parametrize("B,nh,L,P", [(2, 2, 16, 16), (2, 2, 50, 64), (1, 3, 128, 128)])
@pytest.mark.parametrize("relkey", [True, False])
def test_attention_probability_dropout_forward_and_backward(pkg, hip, B, nh, L, P, relkey):
    from e3diff_amd.autograd import functional as F
    ops = pkg.ops
    H, p = nh * 64, 0.1
    qkv = torch.randn(B * L, 3 * H, generator=g(L))
    E = torch.randn(2 * P - 1, 64, generator=g(P)) if relkey else None
    lens = torch.randint(1, L + 1, (B,), generator=g(3))
    lens[0] = L
    mask = (torch.arange(L)[None] < lens[:, None]).float()
    go = torch.randn(B * L, H, generator=g(9))
    sp = lambda x: x.reshape(B, L, nh, 64).permute(0, 2, 1, 3)  # noqa: E731
    for mode, tol in (("bf16x6", 2e-5), ("bf16x3", 1e-4), ("f32", 2e-5)):
        prev = ops.set_attn_mode(mode)
        try:
            torch.manual_seed(100)
            seed = ops.next_dropout_seed()
            mult = ops.attn_dropout_mask(B, nh, L, L, p, seed).cpu()
            vals = torch.unique(mult)
            assert vals.tolist() == [0.0, pytest.approx(65536.0 / (65536 - 6554))]
            assert abs((mult == 0).float().mean().item() - 0.1) < 0.02
            qr = leaf(qkv, dtype=torch.double)
            Er = leaf(E, dtype=torch.double) if relkey else None
            ref = ref_attention_dropout(sp(qr[:, :H]), sp(qr[:, H:2 * H]), sp(qr[:, 2 * H:]), mask.double(), Er, P,
                                        mult.double())
            ref2d = ref.permute(0, 2, 1, 3).reshape(B * L, H)
            ref2d.backward(go.double())
            torch.manual_seed(100)             # F.attention draws the same seed
            qd = leaf(qkv, DEV)
            Ed = leaf(E, DEV) if relkey else None
            out = F.attention(qd, None, B, nh, L, L, key_mask=mask.to(DEV), dist_emb=Ed, max_pos=P, drop_p=p)
            out.backward(go.to(DEV))
        finally:
            ops.set_attn_mode(prev)
        assert rel_err(out, ref2d.float()) < tol, mode
        assert rel_err(qd.grad, qr.grad.float()) < tol, mode
        if relkey:
            assert rel_err(Ed.grad, Er.grad.float()) < tol, mode


def _tiny_structure_model(pkg, p):
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.structure_model.model import ConditionalBertForDiffusion as M
    c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=1,
             max_position_embeddings=32, hidden_dropout_prob=p, attention_probs_dropout_prob=p)
    torch.manual_seed(0)
    return M(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), feature_names=list("abcdefgh"),
             loss_func=[M.diheral_loss_func] * 4 + [M.angle_loss_func] * 4, l2_lambda=0.0).to(DEV)


def test_model_dropout_is_active_in_training_only_and_seedable(pkg, hip):
    from helpers import synthetic_pockets
    from e3diff_amd.structure_model.dataset import noise_batch_on_device
    from e3diff_amd.structure_model.utils import CosineTables
    model = _tiny_structure_model(pkg, 0.1)
    pk = {k: v.to(DEV) for k, v in synthetic_pockets(4, 32, seed=0).items() if torch.is_tensor(v)}
    torch.manual_seed(3)
    batch = dict(pk, **noise_batch_on_device(pk["ligand_angles"], CosineTables(1000)))
    args = (batch["timestep"], batch["noised_ligand_angle"], batch["ligand_attn_mask"], batch["receptor_seq"],
            batch["receptor_angles"], batch["receptor_attn_mask"])
    model.eval()
    with torch.no_grad():
        e1, e2 = model(*args), model(*args)
    assert torch.equal(e1, e2)
    model.train()
    torch.manual_seed(11)
    t1 = model(*args)
    torch.manual_seed(11)
    t2 = model(*args)
    torch.manual_seed(12)
    t3 = model(*args)
    assert torch.equal(t1, t2)                       # repeatable under torch.manual_seed
    assert not torch.equal(t1, t3)                   # different draw
    assert rel_err(t1, e1) > 1e-3                    # and it does something
    t1.square().mean().backward()                    # gradients flow through every dropout site
    grads = [p.grad for n, p in model.named_parameters() if p.grad is not None]
    assert grads and all(torch.isfinite(gr).all() for gr in grads)


@pytest.mark.parametrize("M,H", [(70, 768), (4096, 768), (33, 256)])
def test_dropout_folded_into_the_residual_layernorm_kernels(pkg, hip, M, H):
    """LayerNorm(dropout(x) + r) with the dropout applied INSIDE the LayerNorm kernels: forward and both input gradients
    are bit-identical to the separate dropout launch followed by the plain kernels for the same (p, seed)."""
    from e3diff_amd import autograd as AG, ops
    g = torch.Generator().manual_seed(5)
    x, r = torch.randn(M, H, generator=g).cuda(), torch.randn(M, H, generator=g).cuda()
    ga, be = (1 + 0.1 * torch.randn(H, generator=g)).cuda(), torch.randn(H, generator=g).cuda()
    dy = torch.randn(M, H, generator=g).cuda()
    p, seed = 0.1, 987654321
    out_f, s_f = ops.residual_layernorm(x, r, ga, be, 1e-12, want_s=True, drop=(p, seed))
    out_u, s_u = ops.residual_layernorm(ops.dropout(x, p, seed), r, ga, be, 1e-12, want_s=True)
    assert torch.equal(out_f, out_u) and torch.equal(s_f, s_u)
    ds_f, dg_f, db_f, dsd = AG.layernorm_bwd(dy, s_f, ga, 1e-12, drop=(p, seed))
    ds_u, dg_u, db_u = AG.layernorm_bwd(dy, s_u, ga, 1e-12)
    assert torch.equal(ds_f, ds_u) and torch.equal(dsd, ops.dropout(ds_u, p, seed))
    for a, b in ((dg_f, dg_u), (db_f, db_u)):          # (summed by atomics: order-dependent last bits)
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
    assert not torch.equal(out_f, ops.residual_layernorm(x, r, ga, be, 1e-12))                                # it does drop
