"""Oracle restatement of the reference structure_model hot path (CPU, fp32).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Each function cites the reference lines it
follows; paths are relative to /root/reference/structure_model/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import bert


# ----------------------------------------------------------------------------- schedule / wrap
def cosine_beta_schedule(timesteps, s=8e-3):
    """utils.py:9-18."""
    steps = timesteps + 1
    x = torch.linspace(0, timesteps, steps)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.clip(betas, 0.0001, 0.9999)


def compute_alphas(betas):
    """utils.py:42-59."""
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
    return {
        "betas": betas,
        "alphas": alphas,
        "alphas_cumprod": ac,
        "sqrt_alphas_cumprod": torch.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
        "posterior_variance": betas * (1.0 - ac_prev) / (1.0 - ac),
    }


def modulo_with_wrapped_range(vals, range_min=-np.pi, range_max=np.pi):
    """utils.py:20-40: ((v - min) mod (max - min)) + min with floored modulo."""
    top_end = range_max - range_min
    return (vals - range_min) % top_end + range_min


def radian_l1_loss(inp, target):
    """utils.py:61-76."""
    target = target % (2 * torch.pi)
    inp = inp % (2 * torch.pi)
    d = target - inp
    d = (d + torch.pi) % (2 * torch.pi) - torch.pi
    return torch.mean(torch.abs(d))


def radian_smooth_l1_loss(inp, target, beta=1.0):
    """utils.py:78-109 (circle_penalty=0, the only value the reference uses)."""
    d = modulo_with_wrapped_range(target - inp, -torch.pi, torch.pi)
    abs_d = torch.abs(d)
    return torch.mean(torch.where(abs_d < beta, 0.5 * (d ** 2) / beta, abs_d - 0.5 * beta))


# ----------------------------------------------------------------------------- model blocks
def extend_mask(mask):
    """model.py:226-231: (1 - m) * -10000 -> [B,1,1,L]."""
    return (1.0 - mask[:, None, None, :].type_as(mask)) * -10000.0


def embeddings(sd, prefix, x, eps=1e-12):
    """BertEmbeddings, model.py:111-118 (eval)."""
    return bert.layer_norm(sd, prefix + ".LayerNorm", bert.linear(sd, prefix + ".linear", x), eps)


def fourier_projection(sd, prefix, t):
    """GaussianFourierProjection.forward, model.py:86-98.  ``t`` keeps its dtype (int64 in
    structure sampling) exactly as the reference multiplies it."""
    if t.ndim > 1:
        t = t.squeeze()
    elif t.ndim < 1:
        t = t.unsqueeze(0)
    if t.ndim < 1:  # squeeze of a [1,1] tensor
        t = t.unsqueeze(0)
    w = sd[prefix + ".W"]
    x_proj = t[:, None] * w[None, :] * 2 * torch.pi
    return torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)


def se_layer(sd, prefix, x, c, mask_bias, num_heads, max_pos):
    """SELayer.forward, model.py:53-67 (adaLN-gated attention + MLP; LayerNorms without affine,
    default eps 1e-5)."""
    h = x.shape[-1]
    mod = bert.linear(sd, prefix + ".adaLN_modulation.2",
                      F.silu(bert.linear(sd, prefix + ".adaLN_modulation.0", c)))
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = mod.chunk(6, dim=-1)
    att = bert.attention_block(sd, prefix + ".attn", x, mask_bias, num_heads, max_pos)
    x = x + gate_msa * (F.layer_norm(att, (h,)) * (1 + scale_msa) + shift_msa)
    mlp = bert.linear(sd, prefix + ".mlp.3", F.gelu(bert.linear(sd, prefix + ".mlp.0", x)))
    x = x + gate_mlp * (F.layer_norm(mlp, (h,)) * (1 + scale_mlp) + shift_mlp)
    return x


def predictor(sd, prefix, x, eps=1e-12):
    """AnglesPredictor / AminoAcidPredictor.forward, model.py:149-154."""
    x = F.gelu(bert.linear(sd, prefix + ".dense1", x))
    x = bert.layer_norm(sd, prefix + ".layer_norm", x, eps)
    return bert.linear(sd, prefix + ".dense2", x)


def forward(sd, cfg, timestep, noised_ligand_angles, ligand_attention_masks,
            receptor_seq, receptor_angles, receptor_attention_masks):
    """ConditionalBertForDiffusionBase.forward, model.py:180-215.

    cfg: dict(num_heads=..., max_pos=...).  Returns predicted noise [B,L,8].
    """
    nh, mp = cfg["num_heads"], cfg["max_pos"]
    lig_bias = extend_mask(ligand_attention_masks)
    rec_bias = extend_mask(receptor_attention_masks)
    rec_angles = embeddings(sd, "receptor_angle_emb", receptor_angles)
    rec_seq = embeddings(sd, "receptor_seq_emb", receptor_seq)
    rec = se_layer(sd, "receptor_emb", rec_angles, rec_seq, rec_bias, nh, mp)
    enc = bert.bert_encoder(sd, "encoder", rec, rec_bias, nh, mp)
    lig = embeddings(sd, "ligand_angle_emb", noised_ligand_angles)
    temb = fourier_projection(sd, "timestep_projector", timestep.squeeze(dim=-1)).unsqueeze(1)
    lig = se_layer(sd, "timestep_emb", lig, temb, lig_bias, nh, mp)
    dec = bert.bert_encoder(sd, "decoder", lig, lig_bias, nh, mp, enc, rec_bias)
    return predictor(sd, "angles_predictor", dec)


# ----------------------------------------------------------------------------- sampler
def p_sample(model_fn, ligand_mask, x_t, receptor_seq, receptor_mask, receptor_angle,
             timestep, betas, noise=None):
    """sample.py:55-99.  ``noise`` injects the Gaussian draw (H3: RNG streams differ across
    generators, so parity tests inject it); None draws torch.randn_like."""
    ab = compute_alphas(betas)
    sqrt_recip_alphas = 1.0 / torch.sqrt(ab["alphas"])
    t_unique = torch.unique(timestep)
    assert len(t_unique) == 1, f"Got multiple values for t: {t_unique}"
    t_index = t_unique.item()
    eps_hat = model_fn(timestep, x_t, ligand_mask, receptor_seq, receptor_angle, receptor_mask)
    mean = sqrt_recip_alphas[t_index] * (
        x_t - betas[t_index] * eps_hat / ab["sqrt_one_minus_alphas_cumprod"][t_index])
    if t_index == 0:
        return mean
    if noise is None:
        noise = torch.randn_like(x_t)
    return mean + torch.sqrt(ab["posterior_variance"][t_index]) * noise


def p_sample_loop(model_fn, ligand_mask, x_T, receptor_seq, receptor_mask, receptor_angle,
                  total_timesteps, betas, noises=None, step=1):
    """sample.py:101-144; returns [T/step, B, L, F].  noises: optional [T/step, B, L, F]."""
    b = x_T.shape[0]
    x = x_T
    out = []
    for n, i in enumerate(reversed(range(0, total_timesteps, step))):
        x = p_sample(model_fn, ligand_mask, x, receptor_seq, receptor_mask, receptor_angle,
                     torch.full((b,), i, dtype=torch.long), betas,
                     None if noises is None else noises[n])
        x = modulo_with_wrapped_range(x, -torch.pi, torch.pi)
        out.append(x)
    return torch.stack(out)


def add_noise_by_timestep(v, t_index, alpha_beta_terms, noise):
    """dataset.py:211-228 with the wrapped noise injected (dataset.py:170-185 wraps randn)."""
    a = alpha_beta_terms["sqrt_alphas_cumprod"][t_index]
    s = alpha_beta_terms["sqrt_one_minus_alphas_cumprod"][t_index]
    return modulo_with_wrapped_range(a * v + s * noise, -np.pi, np.pi)


def loss_terms(pred, known_noise, ligand_mask, n_dihedral=4):
    """_get_loss_terms, model.py:266-303 with loss_func = [radian_l1]*4 + [smooth_l1(beta=pi/10)]*4
    (train_model.py:94-95, model.py:238-239)."""
    idx = torch.where(ligand_mask)
    terms = []
    for i in range(known_noise.shape[-1]):
        p, k = pred[idx[0], idx[1], i], known_noise[idx[0], idx[1], i]
        terms.append(radian_l1_loss(p, k) if i < n_dihedral
                     else radian_smooth_l1_loss(p, k, beta=math.pi / 10))
    return torch.stack(terms)
