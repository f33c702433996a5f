"""BERT encoder/decoder stacks with ``relative_key`` self-attention, executed by HIP kernels.

Mirrors the module tree -- and therefore the ``state_dict`` key names, the checkpoint wire
format -- of transformers==4.38.2 ``BertEncoder`` / ``BertLayer`` /
``BertAttention`` as the reference instantiates them (structure_model/model.py:16-20,40,171,177),
but none of that package's code runs: the nn.Modules below only own parameters, and ``run_*``
execute the math as fused gfx950 kernels on [B*L, H] row-major activations.
"""
from dataclasses import dataclass

import os

import torch
from torch import nn

from . import ops
from .autograd import functional as F


@dataclass
class BertConfig:
    """The subset of transformers.BertConfig the reference sets (structure_model/sample.py:160-183).
    Any object with these attributes (e.g. a real transformers.BertConfig) is accepted too."""
    hidden_size: int = 768
    num_attention_heads: int = 12
    intermediate_size: int = 1024
    num_hidden_layers: int = 12
    max_position_embeddings: int = 64
    position_embedding_type: str = "relative_key"
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    layer_norm_eps: float = 1e-12
    hidden_act: str = "gelu"
    use_cache: bool = False
    is_decoder: bool = False
    add_cross_attention: bool = False


def _check_config(cfg):
    head = cfg.hidden_size // cfg.num_attention_heads
    if cfg.hidden_size % cfg.num_attention_heads or head != 64:
        raise ValueError(f"HIP attention kernel needs head dim 64 (hidden {cfg.hidden_size}, "
                         f"heads {cfg.num_attention_heads})")
    if cfg.hidden_size not in (256, 512, 768, 1024):
        raise ValueError("HIP row kernels need hidden_size in {256,512,768,1024}")
    if cfg.intermediate_size % 128:
        raise ValueError("HIP GEMM needs intermediate_size % 128 == 0")
    if getattr(cfg, "hidden_act", "gelu") != "gelu":
        raise ValueError("only the exact-erf GELU the reference uses is implemented")
    if cfg.position_embedding_type not in ("relative_key", "absolute"):
        raise ValueError(f"unsupported position_embedding_type {cfg.position_embedding_type}")


# ----------------------------------------------------------------------------- parameter tree
class BertSelfAttention(nn.Module):
    def __init__(self, cfg, position_embedding_type=None):
        super().__init__()
        h = cfg.hidden_size
        self.query, self.key, self.value = nn.Linear(h, h), nn.Linear(h, h), nn.Linear(h, h)
        self.position_embedding_type = position_embedding_type or cfg.position_embedding_type
        self.max_position_embeddings = cfg.max_position_embeddings
        if self.position_embedding_type == "relative_key":
            self.distance_embedding = nn.Embedding(2 * cfg.max_position_embeddings - 1,
                                                   h // cfg.num_attention_heads)


class BertSelfOutput(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class BertAttention(nn.Module):
    def __init__(self, cfg, position_embedding_type=None):
        super().__init__()
        _check_config(cfg)
        self.self = BertSelfAttention(cfg, position_embedding_type)
        self.output = BertSelfOutput(cfg)
        self.num_heads = cfg.num_attention_heads
        self.eps = cfg.layer_norm_eps
        self.config = cfg


class BertIntermediate(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.intermediate_size)


class BertOutput(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.intermediate_size, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class BertLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.attention = BertAttention(cfg)
        if cfg.add_cross_attention:
            if not cfg.is_decoder:
                raise ValueError("cross attention needs is_decoder=True")
            # 4.38.2 BertLayer builds the cross attention with position_embedding_type="absolute"
            self.crossattention = BertAttention(cfg, position_embedding_type="absolute")
        self.intermediate = BertIntermediate(cfg)
        self.output = BertOutput(cfg)
        self.eps = cfg.layer_norm_eps


class BertEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        _check_config(cfg)
        self.config = cfg
        self.layer = nn.ModuleList([BertLayer(cfg) for _ in range(cfg.num_hidden_layers)])


# ----------------------------------------------------------------------------- packed weights
TIE_PACKED = os.environ.get("E3D_TIE_QKV", "1") == "1"     # 0: a fresh torch.cat per forward (A/B runs)


class _TiedPack(torch.autograd.Function):
    """(packed weight, packed bias) as differentiable functions of the parameters whose storage they ALIAS (``_tie``): no
    copy and no kernel in either direction.  The backward (only reached when the weight gradients are not deferred:
    autograd.deferred_weight_grads writes the row blocks' gradients straight into the parameters) hands each parameter
    its row block of the packed gradient."""

    @staticmethod
    def forward(ctx, w_packed, b_packed, n, *parts):
        ctx.n, ctx.rows = n, [p.shape[0] for p in parts[:n]]
        ctx.set_materialize_grads(False)      # deferred weight gradients hand back None: no zero tensors, no zero accumulation
        return w_packed.view(w_packed.shape), b_packed.view(b_packed.shape)

    @staticmethod
    def backward(ctx, gw, gb):
        gws = gw.split(ctx.rows, 0) if gw is not None else (None,) * ctx.n
        gbs = gb.split(ctx.rows, 0) if gb is not None else (None,) * ctx.n
        return (None, None, None) + tuple(gws) + tuple(gbs)


def _tie(owner, name, tensors):
    """One buffer [sum rows, ...] whose row blocks ARE the given parameters: on first use (and whenever a parameter has
    moved: ``module.to()``, a re-assigned ``.data``) the values are packed once and every parameter's ``.data`` becomes a
    view of its block.  In-place updates (optimizers, ``load_state_dict``, broadcasts) then keep the packed tensor current
    for free; names, shapes and ``state_dict`` are unchanged."""
    tied = owner.__dict__.setdefault("_e3d_tied", {})
    ent = tied.get(name)
    ok = ent is not None and ent.device == tensors[0].device
    if ok:
        at = ent.data_ptr()
        for t in tensors:
            if t.data_ptr() != at or not t.is_contiguous():
                ok = False
                break
            at += 4 * t.numel()
    if not ok:
        with torch.no_grad():
            ent = torch.cat([t.detach() for t in tensors], dim=0).contiguous()
            r0 = 0
            for t in tensors:
                t.data = ent[r0:r0 + t.shape[0]]
                r0 += t.shape[0]
        tied[name] = ent
    return ent


def _packed(owner, name, weights, biases):
    """(cat(weights), cat(biases)) of per-projection parameters.  Inference: copies cached on the owner and rebuilt when
    any source parameter changes (in-place update or re-assignment).  Training (a source requires grad and autograd is
    on): the parameters are tied to packed buffers (``_tie``) and the pair is a zero-copy alias of them -- a training
    step used to spend 72 cat kernels (0.5 ms) re-packing weights the optimizer had just updated."""
    if torch.is_grad_enabled() and any(t.requires_grad for t in weights):
        if TIE_PACKED and all(t.dtype == torch.float32 for t in list(weights) + list(biases)):
            w, b = _TiedPack.apply(_tie(owner, "w_" + name, weights), _tie(owner, "b_" + name, biases), len(weights),
                                   *weights, *biases)
        else:
            w, b = torch.cat(list(weights), dim=0), torch.cat(list(biases), dim=0)
        w._e3d_parts, b._e3d_parts = list(weights), list(biases)   # autograd.deferred_weight_grads writes the row blocks'
        return w, b                                                 # gradients straight into these parameters
    out = []
    cache = owner.__dict__.setdefault("_e3d_pack", {})
    for key_name, tensors in (("w_" + name, weights), ("b_" + name, biases)):
        key = ops.weight_key(*tensors)    # includes the optimizer-step generation: fused optimizers do not bump _version
        hit = cache.get(key_name)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, torch.cat([t.detach() for t in tensors], dim=0).contiguous())
            cache[key_name] = hit
        out.append(hit[1])
    return out[0], out[1]


def qkv_weights(sa):
    return _packed(sa, "qkv", [sa.query.weight, sa.key.weight, sa.value.weight], [sa.query.bias, sa.key.bias, sa.value.bias])


def kv_weights(sa):
    return _packed(sa, "kv", [sa.key.weight, sa.value.weight], [sa.key.bias, sa.value.bias])


def dropout_rates(module):
    """(hidden_dropout_prob, attention_probs_dropout_prob) in training mode, (0, 0) in eval -- where
    transformers 4.38.2 applies ``nn.Dropout``: after the self/cross-attention softmax, after
    ``BertSelfOutput.dense`` and after ``BertOutput.dense``."""
    if not module.training:
        return 0.0, 0.0
    cfg = module.config
    return float(getattr(cfg, "hidden_dropout_prob", 0.0)), float(getattr(cfg, "attention_probs_dropout_prob", 0.0))


# ----------------------------------------------------------------------------- executors
def _attention_output(att, ctx, x, p_hidden=0.0):
    return F.linear_residual_layernorm(ctx, att.output.dense.weight, att.output.dense.bias, x, att.output.LayerNorm.weight,
                                       att.output.LayerNorm.bias, att.eps, p_hidden)


def run_self_attention(att, x, mask, B, L, drop=(0.0, 0.0)):
    """BertAttention on x [B*L,H] with key padding mask [B,L] (1/0): fused QKV GEMM ->
    fused relative-key attention -> out-proj GEMM -> residual + LayerNorm."""
    sa = att.self
    w, b = qkv_weights(sa)
    qkv = F.linear(x, w, b, absmax=ops.absmax_slot(sa, "qkv", x.device))
    relkey = sa.position_embedding_type == "relative_key"
    ctx = F.attention(qkv, None, B, att.num_heads, L, L, key_mask=mask,
                      dist_emb=sa.distance_embedding.weight if relkey else None,
                      max_pos=sa.max_position_embeddings, drop_p=drop[1])
    return _attention_output(att, ctx, x, drop[0])


def project_cross_kv(att, enc):
    """K/V projection of the encoder states for one decoder layer -> [B*Lk, 2H].  It depends on
    neither the timestep nor the noised ligand, so samplers compute it once."""
    w, b = kv_weights(att.self)
    # (a scalar of its own, not a pool slot: samplers keep this projection -- and its bound -- for a whole chain)
    return F.linear(enc, w, b, absmax=torch.zeros(1, device=enc.device, dtype=torch.float32))


def run_cross_attention(att, x, kv, enc_mask, B, Lq, Lk, drop=(0.0, 0.0)):
    q = F.linear(x, att.self.query.weight, att.self.query.bias, absmax=ops.absmax_slot(att.self, "q", x.device))
    ctx = F.attention(q, kv, B, att.num_heads, Lq, Lk, key_mask=enc_mask, drop_p=drop[1])
    return _attention_output(att, ctx, x, drop[0])


def run_layer(layer, x, mask, B, L, cross_kv=None, enc_mask=None, Lk=None, drop=(0.0, 0.0)):
    """``drop`` = (hidden, attention-probability) dropout rates of this call (training only)."""
    x = run_self_attention(layer.attention, x, mask, B, L, drop)
    if hasattr(layer, "crossattention"):
        if cross_kv is None:
            raise ValueError("decoder layer needs encoder states")
        x = run_cross_attention(layer.crossattention, x, cross_kv, enc_mask, B, L, Lk, drop)
    inter = F.linear(x, layer.intermediate.dense.weight, layer.intermediate.dense.bias, ops.ACT_GELU)
    return F.linear_residual_layernorm(inter, layer.output.dense.weight, layer.output.dense.bias, x,
                                       layer.output.LayerNorm.weight, layer.output.LayerNorm.bias, layer.eps, drop[0])


def run_encoder(encoder, x, mask, B, L, enc=None, enc_mask=None, Lk=None, cross_kv=None):
    """BertEncoder(...).last_hidden_state on flat activations.  ``cross_kv`` (list, one per
    layer) short-cuts the per-layer K/V projection of ``enc``."""
    drop = dropout_rates(encoder)
    for i, layer in enumerate(encoder.layer):
        kv = None
        if hasattr(layer, "crossattention"):
            kv = cross_kv[i] if cross_kv is not None else project_cross_kv(layer.crossattention, enc)
        x = run_layer(layer, x, mask, B, L, kv, enc_mask, Lk, drop)
    return x
