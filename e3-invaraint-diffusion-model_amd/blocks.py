"""Blocks shared by the structure and sequence denoisers (the reference duplicates them:
structure_model/model.py:27-154 == sequence_model/model.py:26-153).  Parameter containers keep
the reference attribute names (checkpoint keys); ``run`` methods execute on HIP kernels."""
import math

import torch
from torch import nn

from . import bert, ops
from .autograd import functional as F


class SELayer(nn.Module):
    """adaLN-gated attention + MLP block (structure_model/model.py:27-67)."""

    def __init__(self, bert_config, mlp_ratio=4.0):
        super().__init__()
        h = bert_config.hidden_size
        self.norm1 = nn.LayerNorm(h, elementwise_affine=False)
        self.norm2 = nn.LayerNorm(h, elementwise_affine=False)
        self.adaLN_modulation = nn.Sequential(nn.Linear(h, h, bias=True), nn.SiLU(),
                                              nn.Linear(h, 6 * h, bias=True))
        self.attn = bert.BertAttention(bert_config)
        self.mlp = nn.Sequential(nn.Linear(h, int(h * mlp_ratio)), nn.GELU(),
                                 nn.Dropout(bert_config.hidden_dropout_prob),
                                 nn.Linear(int(h * mlp_ratio), h),
                                 nn.Dropout(bert_config.hidden_dropout_prob))
        nn.init.zeros_(self.adaLN_modulation[0].weight)
        nn.init.zeros_(self.adaLN_modulation[0].bias)

    def modulation(self, c):
        """adaLN_modulation(c): [n,H] -> [n,6H] (shift, scale, gate of the attention branch, then of the MLP branch)."""
        m0, m2 = self.adaLN_modulation[0], self.adaLN_modulation[2]
        return F.linear(F.linear(c, m0.weight, m0.bias, ops.ACT_SILU), m2.weight, m2.bias)

    def run(self, x, c, mask, B, L, mod=None):
        """x [B*L,H]; c [B*L,H] (per token) or [B,H] (one conditioning row per item).  ``mod``: the rows
        ``modulation(c)`` would give, computed by the caller ([B*L,6H], [B,6H], or ONE row [1,6H] shared by every
        item: samplers precompute it per timestep) -- ``c`` is then not read."""
        if mod is None:
            mod = self.modulation(c)
        rows_per_cond = x.shape[0] // mod.shape[0]
        assert rows_per_cond in (1, L, B * L) and rows_per_cond * mod.shape[0] == x.shape[0], (x.shape, mod.shape)
        drop = bert.dropout_rates(self.attn)   # (hidden, attention) rates in training, zeros in eval
        att = bert.run_self_attention(self.attn, x, mask, B, L, drop)
        x = F.adaln_gate(x, att, mod, 0, rows_per_cond)
        h = F.dropout(F.linear(x, self.mlp[0].weight, self.mlp[0].bias, ops.ACT_GELU), self.mlp[2].p, self.training)
        h = F.dropout(F.linear(h, self.mlp[3].weight, self.mlp[3].bias), self.mlp[4].p, self.training)
        return F.adaln_gate(x, h, mod, 1, rows_per_cond)


class GaussianFourierProjection(nn.Module):
    """structure_model/model.py:69-98.  Kept as three tiny torch device ops on purpose: with raw
    integer timesteps the sin/cos arguments reach ~1e5 rad, and the reference's op order
    (t*W, *2, *pi in fp32) fixes which fp32 argument is reduced."""

    def __init__(self, embed_dim=384, scale=2 * math.pi):
        super().__init__()
        self.register_buffer("W", torch.randn(embed_dim // 2) * scale)

    def forward(self, x):
        if x.ndim > 1:
            x = x.squeeze()
        elif x.ndim < 1:
            x = x.unsqueeze(0)
        if x.ndim < 1:
            x = x.unsqueeze(0)
        x_proj = x[:, None] * self.W[None, :] * 2 * torch.pi
        return torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)


class BertEmbeddings(nn.Module):
    """Linear -> LayerNorm -> dropout (structure_model/model.py:100-118)."""

    def __init__(self, in_features, bert_config):
        super().__init__()
        self.linear = nn.Linear(in_features, bert_config.hidden_size)
        self.LayerNorm = nn.LayerNorm(bert_config.hidden_size, eps=bert_config.layer_norm_eps)
        self.dropout = nn.Dropout(bert_config.hidden_dropout_prob)

    def run(self, x2d, post_add=None, rows_per_add=1):
        """``post_add`` [M / rows_per_add, H] is what the caller adds to the embedding afterwards (the sequence
        model's timestep term): fused into the kernel, except in training with dropout, which sits between."""
        if self.training and self.dropout.p > 0:
            e = F.embed_layernorm(x2d, self.linear.weight, self.linear.bias, self.LayerNorm.weight,
                                  self.LayerNorm.bias, self.LayerNorm.eps, None, 1)
            e = F.dropout(e, self.dropout.p)
            if post_add is not None:
                e = (e.view(-1, rows_per_add, e.shape[1]) + post_add[:, None, :]).view_as(e)
            return e
        return F.embed_layernorm(x2d, self.linear.weight, self.linear.bias, self.LayerNorm.weight,
                                   self.LayerNorm.bias, self.LayerNorm.eps, post_add, rows_per_add)


class Predictor(nn.Module):
    """AnglesPredictor / AminoAcidPredictor: dense -> GELU -> LayerNorm -> dense
    (structure_model/model.py:120-154)."""

    def __init__(self, d_model, d_out, eps=1e-12):
        super().__init__()
        self.d_model, self.d_out = d_model, d_out
        self.dense1 = nn.Linear(d_model, d_model)
        self.layer_norm = nn.LayerNorm(d_model, eps=eps)
        self.dense2 = nn.Linear(d_model, d_out)

    def run(self, x):
        h = F.linear(x, self.dense1.weight, self.dense1.bias, ops.ACT_GELU)
        h = F.residual_layernorm(h, None, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        return F.head_linear(h, self.dense2.weight, self.dense2.bias)


def flat2d(x):
    """[B,L,F] -> contiguous fp32 [B*L,F] view for the kernels."""
    return x.reshape(-1, x.shape[-1]).contiguous().float()


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("the denoiser runs on HIP kernels only: move the model and its inputs to a "
                               "GPU device (there is no CPU fallback; the CPU oracle lives in oracle/ and is "
                               "test infrastructure)")
