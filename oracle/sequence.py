"""Oracle restatement of the reference sequence_model hot path (CPU, fp32 + integer indexing).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Paths are relative to
/root/reference/sequence_model/.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import bert
from .structure import embeddings, extend_mask, fourier_projection, predictor, se_layer


# ----------------------------------------------------------------------------- schedule
def cosine_beta_schedule_discrete(timesteps, s=0.008):
    """utils.py:99-108 (numpy float64)."""
    steps = timesteps + 2
    x = np.linspace(0, steps, steps)
    ac = np.cos(0.5 * np.pi * ((x / steps) + s) / (1 + s)) ** 2
    ac = ac / ac[0]
    alphas = ac[1:] / ac[:-1]
    return (1 - alphas).squeeze()


class NoiseScheduleDiscrete:
    """PredefinedNoiseScheduleDiscrete, utils.py:206-233."""

    def __init__(self, timesteps):
        self.timesteps = timesteps
        self.betas = torch.from_numpy(cosine_beta_schedule_discrete(timesteps)).float()
        alphas = 1 - torch.clamp(self.betas, min=0, max=0.9999)
        self.alphas_bar = torch.exp(torch.cumsum(torch.log(alphas), dim=0))

    def t_index(self, t_normalized):
        # torch.round is round-half-to-even -- the INT index must be bit-exact
        return torch.round(t_normalized * self.timesteps).long()

    def get_alpha_bar(self, t_normalized):
        return self.alphas_bar[self.t_index(t_normalized)]


class UniformTransition:
    """DiscreteUniformTransition.get_Qt_bar, utils.py:258-271."""

    def __init__(self, x_classes=20):
        self.k = x_classes

    def get_Qt_bar(self, alpha_bar_t):
        a = alpha_bar_t.unsqueeze(1)
        u = torch.ones(1, self.k, self.k) / self.k
        return a * torch.eye(self.k).unsqueeze(0) + (1 - a) * u


class BlosumTransition:
    """BlosumTransition, utils.py:273-314.  ``blosum`` is the dict of blosum_substitute.pt
    (data fixture).  The 500-long temperature tables are always interpolated to 501 (the ``if``
    at utils.py:286 compares shape[0]==1 with 500)."""

    def __init__(self, blosum, timestep=500):
        self.score = blosum["original_score"]
        self.timestep = timestep
        t = blosum["Qtb_temperature"].view(1, 1, -1)
        self.temperature = F.interpolate(t, size=timestep + 1, mode="linear",
                                         align_corners=True).squeeze()

    def t_index(self, x):
        return torch.round(x * self.timestep).long()

    def get_Qt_bar(self, x):
        """x is alpha-bar (callers pass alpha-bar, not t: model.py:298-299, sample.py:156-159)."""
        temp = self.temperature[self.t_index(x)]            # [B,1]
        q = torch.softmax(self.score.unsqueeze(0) / temp.unsqueeze(2), dim=2)
        q[q < 1e-6] = 1e-6
        return q


# ----------------------------------------------------------------------------- model
def forward(sd, cfg, timestep, noised_ligand_seq, ligand_angle, ligand_attention_masks,
            receptor_seq, receptor_angle, receptor_attention_masks):
    """ConditionalBertForDiffusionBase.forward, model.py:200-237 (quirk kept: the receptor goes
    through ``ligand_feature_emb`` too, model.py:221)."""
    nh, mp = cfg["num_heads"], cfg["max_pos"]
    lig_bias = extend_mask(ligand_attention_masks)
    rec_bias = extend_mask(receptor_attention_masks)
    temb = fourier_projection(sd, "timestep_projector", timestep.squeeze(dim=-1)).unsqueeze(1)
    lig_seq = embeddings(sd, "ligand_seq_embedding", noised_ligand_seq)
    lig_ang = embeddings(sd, "ligand_angle_embedding", ligand_angle) + temb
    lig = se_layer(sd, "ligand_feature_emb", lig_seq, lig_ang, lig_bias, nh, mp)
    rec_seq = embeddings(sd, "receptor_seq_embedding", receptor_seq)
    rec_ang = embeddings(sd, "receptor_angle_embedding", receptor_angle) + temb
    rec = se_layer(sd, "ligand_feature_emb", rec_seq, rec_ang, rec_bias, nh, mp)
    dec = bert.bert_encoder(sd, "decoder", lig, lig_bias, nh, mp, rec, rec_bias)
    dec = se_layer(sd, "decoder_normalize", dec, temb, lig_bias, nh, mp)
    return predictor(sd, "amino_acid_predictor", dec)


# ----------------------------------------------------------------------------- forward noising
def aa_noise_prob(ligand_seq, t_int, schedule, transition):
    """apply_aa_noise up to the sampling loop, model.py:291-300: prob[N,20] = Qtb[b] @ onehot."""
    b, l, c = ligand_seq.shape
    flat = ligand_seq.reshape(b * l, c)
    rep = torch.arange(b).repeat_interleave(l)
    qtb = transition.get_Qt_bar(schedule.get_alpha_bar(t_int / schedule.timesteps))
    return (qtb[rep] @ flat.unsqueeze(2)).squeeze(-1)


def categorical_from_uniform(prob, u):
    """Inverse-CDF draw used by the parity tests in place of torch.multinomial (H3: RNG streams
    cannot match): index = #{k : cumsum(prob)[k] <= u * sum(prob)}, clamped to the last class
    with prob > 0... kept simple: clamped to C-1.  prob [N,C], u [N] in [0,1)."""
    cdf = torch.cumsum(prob, dim=-1)
    thr = (u * cdf[:, -1]).unsqueeze(-1)
    return torch.clamp((cdf <= thr).sum(dim=-1), max=prob.shape[-1] - 1)


def apply_aa_noise(ligand_seq, t_int, schedule, transition, u=None):
    """model.py:291-311; zero-probability (padding) rows -> class 0."""
    b, l, c = ligand_seq.shape
    prob = aa_noise_prob(ligand_seq, t_int, schedule, transition)
    if u is None:
        idx = torch.stack([p.multinomial(1)[0] if p.sum() != 0 else torch.tensor(0) for p in prob])
    else:
        idx = categorical_from_uniform(prob, u.reshape(-1))
        idx = torch.where(prob.sum(-1) != 0, idx, torch.zeros_like(idx))
    return F.one_hot(idx.reshape(b, l), num_classes=c).float()


# ----------------------------------------------------------------------------- reverse sampler
def posterior_over0(x_t, q_t, qsb, qtb, batch):
    """compute_batched_over0_posterior_distribution, sample.py:120-139."""
    left = x_t.unsqueeze(-2) @ q_t.transpose(-1, -2)[batch]      # [N,1,C]
    num = left * qsb[batch]                                       # [N,C,C]
    den = qtb[batch] @ x_t.unsqueeze(2)                           # [N,C,1]
    den[den == 0] = 1e-6
    return num / den


def reverse_prob(t, s, noised_data, logits, schedule, transition):
    """sample_p_zs_given_zt_discrete up to the sampling loop, sample.py:149-168 -> prob_X [N,C]."""
    b, l, c = noised_data.shape
    rep = torch.arange(b).repeat_interleave(l)
    x_t = noised_data.reshape(b * l, c)
    qtb = transition.get_Qt_bar(schedule.get_alpha_bar(t))
    qsb = transition.get_Qt_bar(schedule.get_alpha_bar(s))
    ratio = qsb / qtb
    q_t = ratio / ratio.sum(dim=-1).unsqueeze(2)
    pred = F.softmax(logits.reshape(b * l, c), dim=-1)
    post = posterior_over0(x_t, q_t, qsb, qtb, rep)
    unnorm = (pred.unsqueeze(-1) * post).sum(dim=1)
    unnorm[torch.sum(unnorm, dim=-1) == 0] = 1e-5
    return unnorm / torch.sum(unnorm, dim=-1, keepdim=True)


def sample_p_zs_given_zt_discrete(t, s, noised_data, logits, schedule, transition, diverse,
                                  is_last_step, u=None):
    """sample.py:141-179.  ``u`` [B*L] uniforms replace multinomial for parity tests."""
    if is_last_step:
        return logits
    b, l, c = noised_data.shape
    prob = reverse_prob(t, s, noised_data, logits, schedule, transition)
    nonzero = prob.sum(-1) != 0
    if diverse:
        if u is None:
            idx = torch.stack([p.multinomial(1)[0] for p in prob])
        else:
            idx = categorical_from_uniform(prob, u.reshape(-1))
    else:
        idx = prob.argmax(dim=-1)
    idx = torch.where(nonzero, idx, torch.zeros_like(idx))
    return F.one_hot(idx.reshape(b, l), num_classes=c).float()


def denoise(model_fn, batch, schedule, transition, diverse, timesteps, x_T, us=None):
    """denoise loop, sample.py:181-207; the model is fed the RAW integer step s (quirk,
    sample.py:194-200).  x_T: initial one-hot [B,L,C]; us: optional [T,B*L] uniforms."""
    b = x_T.shape[0]
    x = x_T
    for n, s_int in enumerate(reversed(range(timesteps))):
        s_array = s_int * torch.ones((b, 1))
        t_array = s_array + 1
        logits = model_fn(s_array, x, batch["ligand_angles"], batch["ligand_attn_mask"],
                          batch["receptor_seq"], batch["receptor_angles"],
                          batch["receptor_attn_mask"])
        x = sample_p_zs_given_zt_discrete(t_array / timesteps, s_array / timesteps, x, logits,
                                          schedule, transition, diverse, s_int == 0,
                                          None if us is None else us[n])
    return x


# ----------------------------------------------------------------------------- losses
def elbo_loss(logits1, logits2, eps=1e-6):
    """utils.py:132-161."""
    probs1 = F.softmax(logits1, dim=-1)
    probs2 = F.softmax(logits2, dim=-1)
    lp1 = F.log_softmax(logits1 + eps, dim=-1)
    kl = F.kl_div(lp1, probs2, reduction="batchmean")
    nll = -torch.mean(torch.sum(probs1 * lp1, dim=-1))
    return nll + kl


def get_loss(pred_aa, batch, noised_ligand_seq):
    """PeptideDiff.get_loss after the forward, model.py:313-345 -> (total, elbo, noised CE, all CE)."""
    ligand_mask = batch["ligand_attn_mask"].bool()
    true_idx = batch["ligand_seq"].argmax(dim=-1)
    noised_mask = noised_ligand_seq.argmax(dim=-1) != true_idx
    ce = torch.nn.CrossEntropyLoss()
    aa_noised = ce(pred_aa[noised_mask].view(-1, 20), true_idx[noised_mask].view(-1))
    keep = ligand_mask & (~noised_mask)
    aa_all = ce(pred_aa[keep].view(-1, 20), true_idx[keep].view(-1))
    elbo = elbo_loss(pred_aa[noised_mask], batch["ligand_seq"][noised_mask])
    return aa_noised + elbo, elbo, aa_noised, aa_all
