"""Oracle restatement of the transformers==4.38.2 BERT blocks the reference calls.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional style: every function takes the
reference ``state_dict`` (``sd``) and a key prefix, so the checkpoint key names are the
contract (SURVEY.md section 8(b) "Checkpoint format").

Third-party algorithm: ``transformers`` pinned 4.38.2 (reference environment.yml:229), not
vendored in /root/reference.  Reference call sites: structure_model/model.py:16-20,40,171,177;
sequence_model/model.py:10-14,39,178.  The ``relative_key`` branch is "parity unpinned" against 4.38.2 itself (not
installed, not vendored).  Independent evidence since round 3: the installed transformers 5.x still ships one
implementation of HuggingFace's relative_key attention (``Wav2Vec2BertSelfAttention``: distance r - l, clamped);
``self_attention`` below reproduces its output to 1e-12 in fp64 when handed the mirrored distance table
(tests/test_oracle_golden.py::test_relative_key_attention_against_an_independent_published_implementation) -- i.e.
everything of SURVEY App. A steps 1-4 except the sign convention of the distance (l - r in 4.38.2 BertSelfAttention).
"""
import math

import torch
import torch.nn.functional as F


def linear(sd, prefix, x):
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


def layer_norm(sd, prefix, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def split_heads(x, num_heads):
    # [B, L, H] -> [B, nh, L, d]   (4.38.2 BertSelfAttention.transpose_for_scores)
    b, l, h = x.shape
    return x.view(b, l, num_heads, h // num_heads).permute(0, 2, 1, 3)


def relkey_scores_literal(q, dist_emb, max_pos):
    """Literal transcription of the 4.38.2 relative_key term.

    q [B,nh,Lq,d]; dist_emb [2P-1,d].  positional = E[l - r + P - 1];
    R = einsum("bhld,lrd->bhlr", q, positional).  Lk == Lq (self-attention only).
    """
    lq = q.shape[2]
    pos_l = torch.arange(lq, dtype=torch.long).view(-1, 1)
    pos_r = torch.arange(lq, dtype=torch.long).view(1, -1)
    distance = pos_l - pos_r
    positional = dist_emb[distance + max_pos - 1].to(q.dtype)  # [L, L, d]
    return torch.einsum("bhld,lrd->bhlr", q, positional)


def self_attention(sd, prefix, hidden, mask_bias, num_heads, max_pos,
                   encoder_hidden=None, encoder_mask_bias=None):
    """4.38.2 BertSelfAttention.forward (eval mode: dropout is identity).

    Self-attention uses ``relative_key`` when ``prefix.distance_embedding.weight`` exists
    (every config of the reference sets position_embedding_type="relative_key":
    structure_model/train_model.py:28).  Cross-attention (``encoder_hidden`` given) is built by
    BertLayer with position_embedding_type="absolute" => no relative term, K/V from the encoder
    states and the encoder mask.
    """
    q = split_heads(linear(sd, prefix + ".query", hidden), num_heads)
    is_cross = encoder_hidden is not None
    kv_src = encoder_hidden if is_cross else hidden
    bias = encoder_mask_bias if is_cross else mask_bias
    k = split_heads(linear(sd, prefix + ".key", kv_src), num_heads)
    v = split_heads(linear(sd, prefix + ".value", kv_src), num_heads)
    scores = torch.matmul(q, k.transpose(-1, -2))
    e_key = prefix + ".distance_embedding.weight"
    if not is_cross and e_key in sd:
        scores = scores + relkey_scores_literal(q, sd[e_key], max_pos)
    scores = scores / math.sqrt(q.shape[-1])
    if bias is not None:
        scores = scores + bias
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(probs, v)
    ctx = ctx.permute(0, 2, 1, 3).contiguous()
    return ctx.view(ctx.shape[0], ctx.shape[1], -1)


def attention_block(sd, prefix, hidden, mask_bias, num_heads, max_pos,
                    encoder_hidden=None, encoder_mask_bias=None, eps=1e-12):
    """BertAttention = BertSelfAttention + BertSelfOutput (LN(dense(ctx) + x))."""
    ctx = self_attention(sd, prefix + ".self", hidden, mask_bias, num_heads, max_pos,
                         encoder_hidden, encoder_mask_bias)
    out = linear(sd, prefix + ".output.dense", ctx)
    return layer_norm(sd, prefix + ".output.LayerNorm", out + hidden, eps)


def bert_layer(sd, prefix, hidden, mask_bias, num_heads, max_pos,
               encoder_hidden=None, encoder_mask_bias=None, eps=1e-12):
    """4.38.2 BertLayer.forward: self-attn -> [cross-attn] -> FFN (post-LN residuals)."""
    x = attention_block(sd, prefix + ".attention", hidden, mask_bias, num_heads, max_pos, eps=eps)
    if (prefix + ".crossattention.self.query.weight") in sd:
        assert encoder_hidden is not None
        x = attention_block(sd, prefix + ".crossattention", x, mask_bias, num_heads, max_pos,
                            encoder_hidden, encoder_mask_bias, eps=eps)
    inter = F.gelu(linear(sd, prefix + ".intermediate.dense", x))  # exact erf GELU
    out = linear(sd, prefix + ".output.dense", inter)
    return layer_norm(sd, prefix + ".output.LayerNorm", out + x, eps)


def bert_encoder(sd, prefix, hidden, mask_bias, num_heads, max_pos,
                 encoder_hidden=None, encoder_mask_bias=None):
    """BertEncoder.forward(...).last_hidden_state; the decoder self-attention is NOT causal
    (the reference calls BertEncoder directly: structure_model/model.py:208-213)."""
    i = 0
    while (prefix + f".layer.{i}.attention.self.query.weight") in sd:
        hidden = bert_layer(sd, prefix + f".layer.{i}", hidden, mask_bias, num_heads, max_pos,
                            encoder_hidden, encoder_mask_bias)
        i += 1
    assert i > 0, f"no layers under {prefix}"
    return hidden
