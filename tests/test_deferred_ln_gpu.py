"""GPU: the deferred-LayerNorm path (inference at large M: include/e3d_hip.h "deferred LayerNorm", bert._run_layer_deferred)
against the classic path -- the same kernels with every LayerNorm as its own pass -- and against fp64."""
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("mode,tol", [("f16x3", 3e-6), ("bf16x3", 6e-5)])
@pytest.mark.parametrize("M,N,K", [(512, 768, 768), (1024, 256, 1024), (256, 1024, 768)])
def test_consumer_and_producer_gemms_against_fp64(pkg, hip, mode, tol, M, N, K):
    """Consumer: act(LayerNorm(z) W^T + b) from the raw z, its row statistics and the folded weight; producer:
    a W^T + b + LayerNorm(z_prev) with the residual rebuilt in the epilogue -- both against fp64, with row means of
    several sigma and an outlier channel in z; the statistics kernel against torch."""
    ops = pkg.ops
    z = torch.randn(M, K, generator=g(1)) * (1 + 3 * torch.rand(M, 1, generator=g(2))) + 2.0 * torch.randn(M, 1, generator=g(3))
    z[:, 5] *= 20
    gamma, beta = 1 + 0.1 * torch.randn(K, generator=g(4)), 0.1 * torch.randn(K, generator=g(5))
    w, b = torch.randn(N, K, generator=g(6)) / K ** 0.5, 0.1 * torch.randn(N, generator=g(7))
    zd, gd, bd, wd, bbd = (t.to(DEV) for t in (z, gamma, beta, w, b))
    stats = ops.row_stats(zd, 1e-12)
    z64 = z.double()
    mu, var = z64.mean(1, keepdim=True), z64.var(1, unbiased=False, keepdim=True)
    assert rel_err(stats[:, 0], mu.squeeze(1).float()) < 1e-5 and rel_err(stats[:, 1], torch.rsqrt(var + 1e-12).squeeze(1).float()) < 1e-5
    h64 = (z64 - mu) * torch.rsqrt(var + 1e-12) * gamma.double() + beta.double()
    with ops.arithmetic(mode, respect_env=False):
        assert ops.gemm_ln_supported(M, N, K, zd)
        wf, bf = ops.folded_linear(wd, bbd, gd, bd)
        for act, fn in ((ops.ACT_NONE, lambda t: t), (ops.ACT_GELU, torch.nn.functional.gelu)):
            got = ops.gemm_ln(zd, wf, bf, act, a_stats=stats)
            assert rel_err(got, fn(h64 @ w.double().t() + b.double()).float()) < tol, (act,)
            assert rel_err(got, ops.gemm(ops.layernorm_from_stats(zd, stats, gd, bd), wd, bbd, act)) < tol     # the classic pair
        # producer: the residual has N columns
        a = torch.randn(M, K, generator=g(8))
        zp = torch.randn(M, N, generator=g(9)) * 3 + torch.randn(M, 1, generator=g(10))
        gp, bp = 1 + 0.1 * torch.randn(N, generator=g(11)), 0.1 * torch.randn(N, generator=g(12))
        zpd = zp.to(DEV)
        sp = ops.row_stats(zpd, 1e-12)
        amax = torch.zeros(1, device=DEV)
        got = ops.gemm_ln(a.to(DEV), wd, bbd, absmax=amax, res=zpd, res_stats=sp, res_gamma=gp.to(DEV), res_beta=bp.to(DEV))
        zp64 = zp.double()
        hp = (zp64 - zp64.mean(1, keepdim=True)) * torch.rsqrt(zp64.var(1, unbiased=False, keepdim=True) + 1e-12) * gp.double() + bp.double()
        want = a.double() @ w.double().t() + b.double() + hp
        assert rel_err(got, want.float()) < tol
        assert float(amax) == float(got.abs().max())


@pytest.mark.parametrize("mode,tol", [("f16x3", 4e-6), ("bf16x3", 1e-4)])
def test_deferred_stack_matches_the_classic_stack(pkg, hip, mode, tol, monkeypatch):
    """run_encoder on an encoder and a cross-attending decoder (relative_key, 3 layers, M = 512 rows): LayerNorm deferred
    through the stack against one LayerNorm pass per site -- same weights, same inputs, padded keys."""
    from e3diff_amd import bert, ops
    cfg = dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=3, max_position_embeddings=128,
               position_embedding_type="relative_key")
    torch.manual_seed(0)
    enc = bert.BertEncoder(bert.BertConfig(**cfg)).eval().to(DEV)
    dec = bert.BertEncoder(bert.BertConfig(**cfg, is_decoder=True, add_cross_attention=True)).eval().to(DEV)
    with torch.no_grad():       # LayerNorm affine parameters away from (1, 0)
        for m in list(enc.modules()) + list(dec.modules()):
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.add_(0.2 * torch.randn_like(m.weight))
                m.bias.add_(0.2 * torch.randn_like(m.bias))
    B, L = 4, 128
    x = torch.randn(B * L, 256, device=DEV)
    y = torch.randn(B * L, 256, device=DEV)
    lens = torch.tensor([128, 70, 33, 5], device=DEV)
    mask = (torch.arange(L, device=DEV)[None] < lens[:, None]).float().contiguous()
    outs = {}
    with torch.no_grad(), ops.arithmetic(mode, respect_env=False):
        for deferred in (True, False):
            monkeypatch.setattr(bert, "DEFER_LN", deferred)
            assert bert._deferred_ok(enc, x) == deferred
            e = bert.run_encoder(enc, x, mask, B, L)
            d = bert.run_encoder(dec, y, mask, B, L, enc=e, enc_mask=mask, Lk=L)
            outs[deferred] = (e.clone(), d.clone())
    assert rel_err(outs[True][0], outs[False][0]) < tol
    assert rel_err(outs[True][1], outs[False][1]) < tol
