"""Shared test helpers: deterministic weights and synthetic BioLiP-shaped inputs.

Nothing here reads /root/reference: the GPU box does not have it.
"""
import math
import os

import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")

TINY = dict(hidden_size=256, num_heads=4, intermediate_size=512, num_hidden_layers=2,
            max_seq_len=16)
FULL_STRUCT = dict(hidden_size=768, num_heads=12, intermediate_size=1024, num_hidden_layers=12)
FULL_SEQ = dict(hidden_size=768, num_heads=12, intermediate_size=1024, num_hidden_layers=6)


def seeded_state_dict(shapes, seed=0, zero_relkey=False):
    """Deterministic weights from (key -> shape), independent of any module's own init.

    Linear weights ~ N(0, 1/fan_in), biases ~ 0.1 N(0,1), LayerNorm gamma = 1 + 0.1 N(0,1),
    Fourier buffer W ~ 2*pi*N(0,1) (structure_model/model.py:82), distance_embedding ~ N(0,1)
    (nn.Embedding default) or zeros when ``zero_relkey``.
    The same function feeds the reference (fixture generation) and product + oracle (tests).
    """
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key in sorted(shapes):
        shape = tuple(shapes[key])
        r = torch.randn(shape, generator=g, dtype=torch.float32)
        if key.endswith("distance_embedding.weight"):
            v = torch.zeros(shape) if zero_relkey else r
        elif key.endswith("timestep_projector.W"):
            v = r * 2 * math.pi
        elif "LayerNorm.weight" in key or "layer_norm.weight" in key:
            v = 1.0 + 0.1 * r
        elif key.endswith(".bias"):
            v = 0.1 * r
        elif len(shape) == 2:
            v = r / math.sqrt(shape[1])
        else:
            v = r
        sd[key] = v.contiguous()
    return sd


def synthetic_pockets(batch, max_len, seed=0, lig_range=(5, 30), rec_range=(20, None),
                      with_ligand_seq=False):
    """Synthetic BioLiP-shaped padded batch in the dataset.py tensor layout
    (structure_model/dataset.py:119-132): dihedrals ~ U(-pi,pi), bond angles ~ N(1.95,0.1),
    uniform one-hot residues, zero padding, float masks."""
    g = torch.Generator().manual_seed(seed)
    rec_hi = rec_range[1] or max_len
    lig_hi = min(lig_range[1], max_len)
    lig_len = torch.randint(min(lig_range[0], lig_hi), lig_hi + 1, (batch,), generator=g)
    rec_len = torch.randint(min(rec_range[0], rec_hi), rec_hi + 1, (batch,), generator=g)

    def angles(lengths):
        a = torch.empty(batch, max_len, 8)
        a[..., :4] = (torch.rand(batch, max_len, 4, generator=g) * 2 - 1) * math.pi
        a[..., 4:] = 1.95 + 0.1 * torch.randn(batch, max_len, 4, generator=g)
        m = (torch.arange(max_len)[None, :] < lengths[:, None]).float()
        return a * m[..., None], m

    def onehot(mask):
        idx = torch.randint(0, 20, (batch, max_len), generator=g)
        return torch.nn.functional.one_hot(idx, 20).float() * mask[..., None]

    lig_angles, lig_mask = angles(lig_len)
    rec_angles, rec_mask = angles(rec_len)
    out = {
        "ligand_angles": lig_angles, "ligand_attn_mask": lig_mask,
        "receptor_angles": rec_angles, "receptor_attn_mask": rec_mask,
        "receptor_seq": onehot(rec_mask),
        "ligand_length": lig_len, "receptor_length": rec_len,
    }
    if with_ligand_seq:
        out["ligand_seq"] = onehot(lig_mask)
    return out


def rel_err(a, b):
    """max |a-b| / max(|b|max, tiny): the "relative fp32" figure the tests assert on."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def reverse_step_tolerance(betas, t_index, eps_scale=1.0, rel=1e-4):
    """Absolute tolerance for ONE reverse step x_t -> x_{t-1} given the north star's ``rel``
    (1e-4 relative fp32) on the predicted noise.  The update is
    sqrt(1/alpha_t) * (x_t - beta_t * eps / sqrt(1 - abar_t)), so an error d on eps is amplified by
    sqrt(1/alpha_t) * beta_t / sqrt(1 - abar_t) -- x100 at t = T-1, where the cosine schedule's
    beta is clipped to 0.9999 (structure_model/utils.py:18).  Free-running trajectories are
    therefore compared step by step (teacher-forced), never end to end."""
    alphas = 1.0 - betas
    abar = torch.cumprod(alphas, 0)
    amp = (1.0 / torch.sqrt(alphas[t_index])) * betas[t_index] / torch.sqrt(1.0 - abar[t_index])
    return rel * eps_scale * float(amp) + 2e-5


def elementwise_err(a, b, floor=1e-3):
    """Element-wise relative-or-absolute error |a-b| / max(|b|, floor * rms(b)) -- the stricter companion of
    ``rel_err`` (which divides every element by the LARGEST reference value).  Returns (99.9th percentile, max)."""
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    rms = b.pow(2).mean().sqrt().clamp_min(1e-30)
    e = (a - b).abs() / torch.maximum(b.abs(), floor * rms)
    k = max(1, int(round(0.999 * e.numel())))
    return e.kthvalue(k).values.item(), e.max().item()


def rescaled_state_dict(sd, weight_scale=4.0, gamma_range=(0.5, 2.0), seed=0):
    """A second weight regime for the arithmetic-margin tests: every Linear weight x ``weight_scale`` (attention
    logits x scale^2: peaky softmax rows), LayerNorm gammas spread log-uniformly over ``gamma_range``."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    lo, hi = math.log(gamma_range[0]), math.log(gamma_range[1])
    for k, v in sd.items():
        if "LayerNorm.weight" in k or "layer_norm.weight" in k:
            out[k] = torch.exp(lo + (hi - lo) * torch.rand(v.shape, generator=g))
        elif v.dim() == 2 and k.endswith(".weight") and "distance_embedding" not in k:
            out[k] = v * weight_scale
        else:
            out[k] = v.clone()
    return out


def heavy_tailed_state_dict(sd, seed=0):
    """A third weight regime for the arithmetic-margin tests, shaped like trained transformer weights rather than like an
    initialiser: every Linear weight row gets its own log-normal scale (sigma 0.5: rows differ by up to ~5x), 0.5 % of the
    entries are outliers (x8), biases x3, LayerNorm gammas log-uniform in [0.3, 3] with 1 % of the channels at x6 (the
    "massive activation" channels of trained LayerNorms), distance tables x0.5."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        if "LayerNorm.weight" in k or "layer_norm.weight" in k:
            gam = torch.exp(math.log(0.3) + (math.log(3.0) - math.log(0.3)) * torch.rand(v.shape, generator=g))
            gam = torch.where(torch.rand(v.shape, generator=g) < 0.01, gam * 6.0, gam)
            out[k] = gam
        elif k.endswith("distance_embedding.weight"):
            out[k] = v * 0.5
        elif v.dim() == 2 and k.endswith(".weight"):
            rows = torch.exp(0.5 * torch.randn(v.shape[0], 1, generator=g))
            w = v * rows
            w = torch.where(torch.rand(v.shape, generator=g) < 0.005, w * 8.0, w)
            out[k] = w.contiguous()
        elif k.endswith(".bias") and v.dim() == 1:
            out[k] = v * 3.0
        else:
            out[k] = v.clone()
    return out
