"""Discrete-diffusion schedule and transition matrices (BLOSUM / uniform) plus the ELBO term.

Public names and call signatures follow the reference's sequence_model/utils.py; results are
checked bit-for-bit against reference-generated fixtures (the table INDEX, a round-half-even of
alpha-bar * 500, must be exact -- BASELINE.json north_star).  Only the live parts are provided:
the reference's dead code (PredefinedNoiseSchedule, custom_beta_schedule_discrete, sigma/alpha
helpers, the duplicated angle losses) is out of scope.
"""
import os

import numpy as np
import torch
from torch.nn import functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))


def cosine_beta_schedule_discrete(timesteps, s=0.008):
    """betas of length timesteps+1 from the squared-cosine alpha-bar curve, float64 numpy
    (reference utils.py:99-108)."""
    n = timesteps + 2
    grid = np.linspace(0, n, n)
    curve = np.cos(0.5 * np.pi * ((grid / n) + s) / (1 + s)) ** 2
    curve = curve / curve[0]
    return (1 - curve[1:] / curve[:-1]).squeeze()


class PredefinedNoiseScheduleDiscrete(torch.nn.Module):
    """Lookup tables beta_t and alpha-bar_t, t = 0..T (reference utils.py:206-233).  Lookups take
    a normalised time and index with torch.round (half-to-even)."""

    def __init__(self, noise_schedule, timesteps):
        super().__init__()
        self.timesteps = timesteps
        self.register_buffer("betas", torch.from_numpy(cosine_beta_schedule_discrete(timesteps)).float())
        self.alphas = 1 - torch.clamp(self.betas, min=0, max=0.9999)
        self.alphas_bar = torch.exp(torch.cumsum(torch.log(self.alphas), dim=0))

    def _index(self, t_normalized, t_int):
        assert int(t_normalized is None) + int(t_int is None) == 1
        if t_int is None:
            t_int = torch.round(t_normalized * self.timesteps)
        return t_int.long()

    def forward(self, t_normalized=None, t_int=None):
        return self.betas[self._index(t_normalized, t_int)]

    def get_alpha_bar(self, t_normalized=None, t_int=None):
        idx = self._index(t_normalized, t_int)
        if self.alphas_bar.device != idx.device:      # moved once (a per-call .to() is a synchronising copy)
            self.alphas_bar = self.alphas_bar.to(idx.device)
        return self.alphas_bar[idx]


class DiscreteUniformTransition:
    """Q = a * I + (1 - a) / K (reference utils.py:235-271)."""

    def __init__(self, x_classes: int):
        self.X_classes = x_classes
        self.u_x = torch.ones(1, x_classes, x_classes)
        if x_classes > 0:
            self.u_x = self.u_x / x_classes

    def _mix(self, keep, device):
        keep = keep.unsqueeze(1).to(device)
        self.u_x = self.u_x.to(device)
        eye = torch.eye(self.X_classes, device=device).unsqueeze(0)
        return keep, eye

    def get_Qt(self, beta_t, device):
        b, eye = self._mix(beta_t, device)
        return b * self.u_x + (1 - b) * eye

    def get_Qt_bar(self, alpha_bar_t, device):
        a, eye = self._mix(alpha_bar_t, device)
        return a * eye + (1 - a) * self.u_x


class BlosumTransition:
    """softmax(BLOSUM score / temperature[idx]) with idx = round(x * 500) (reference
    utils.py:273-314).  The two 500-entry temperature tables are linearly interpolated to 501
    entries at construction (the reference's size test always takes that branch).  Callers pass
    alpha-bar as ``t_normal`` (reference model.py:298-299, sample.py:156-159): kept as is."""

    def __init__(self, blosum_path="./blosum_substitute.pt", x_classes=20, timestep=500):
        tables = None
        for cand in (blosum_path, os.path.join("..", blosum_path),
                     os.path.join(_HERE, os.path.basename(blosum_path))):
            if os.path.exists(cand):
                tables = torch.load(cand, weights_only=True)
                break
        if tables is None:
            raise FileNotFoundError(blosum_path)
        self.original_score = tables["original_score"]
        self.X_classes, self.timestep = x_classes, timestep

        def stretch(v):
            return F.interpolate(v.view(1, 1, -1), size=timestep + 1, mode="linear", align_corners=True).squeeze()

        self.temperature_list = stretch(tables["Qtb_temperature"])
        self.Qt_temperature = stretch(tables["Qt_temperature"])

    def table_index(self, t_normal):
        """The integer lookup the north star requires bit-exact."""
        return torch.round(t_normal * self.timestep).long()

    def _softmax_at(self, table, t_normal, device):
        self.original_score = self.original_score.to(device)
        temp = table.to(device)[self.table_index(t_normal).to(device)]
        return torch.softmax(self.original_score.unsqueeze(0) / temp.unsqueeze(2), dim=2)

    def get_Qt_bar(self, t_normal, device):
        self.temperature_list = self.temperature_list.to(device)
        q_x = self._softmax_at(self.temperature_list, t_normal, device)
        q_x[q_x < 1e-6] = 1e-6
        return q_x

    def get_Qt(self, t_normal, device):
        self.Qt_temperature = self.Qt_temperature.to(device)
        return self._softmax_at(self.Qt_temperature, t_normal, device)


def elbo_loss(logits1, logits2, eps=1e-6):
    """-mean sum p log p  +  KL_batchmean(log_softmax(logits1 + eps) || softmax(logits2))
    (reference utils.py:132-161; ``logits2`` is the one-hot target in get_loss)."""
    p_model = F.softmax(logits1, dim=-1)
    p_target = F.softmax(logits2, dim=-1)
    logp_model = F.log_softmax(logits1 + eps, dim=-1)
    kl = F.kl_div(logp_model, p_target, reduction="batchmean")
    entropy = -torch.mean(torch.sum(p_model * logp_model, dim=-1))
    return entropy + kl
