"""Host-side tables and angle arithmetic of the continuous (cosine) diffusion.

Public names follow the reference's structure_model/utils.py (same call signatures and
results, checked bit-for-bit against reference-generated fixtures); the per-step device work
of the sampler does not go through here -- see sample.py, which runs the HIP
``e3d_ddpm_step_wrap`` kernel.
"""
import math

import numpy as np
import torch

TWO_PI = 2 * math.pi


class CosineTables:
    """All per-timestep scalars of the DDPM, built once on the host as fp32 tensors.

    reference: cosine_beta_schedule (utils.py:9-18) + compute_alphas (utils.py:42-59), which the
    reference re-derives on every reverse step (sample.py:74).
    """

    def __init__(self, timesteps: int, s: float = 8e-3):
        grid = torch.linspace(0, timesteps, timesteps + 1)
        f = torch.cos(((grid / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
        f = f / f[0]
        self.timesteps = timesteps
        self.betas = torch.clip(1 - (f[1:] / f[:-1]), 0.0001, 0.9999)
        self._derive()

    @classmethod
    def from_betas(cls, betas: torch.Tensor) -> "CosineTables":
        self = cls.__new__(cls)
        self.timesteps = betas.shape[0]
        self.betas = betas
        self._derive()
        return self

    def _derive(self):
        b = self.betas
        self.alphas = 1.0 - b
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        prev = torch.cat([torch.ones(1, dtype=b.dtype), self.alphas_cumprod[:-1]])
        self.posterior_variance = b * (1.0 - prev) / (1.0 - self.alphas_cumprod)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = torch.sqrt(1.0 - self.alphas_cumprod)
        # reverse-step scalars (sample.py:75,97)
        self.sqrt_recip_alphas = 1.0 / torch.sqrt(self.alphas)
        self.sigma = torch.sqrt(self.posterior_variance)

    def as_dict(self):
        keys = ("betas", "alphas", "alphas_cumprod", "sqrt_alphas_cumprod",
                "sqrt_one_minus_alphas_cumprod", "posterior_variance")
        return {k: getattr(self, k) for k in keys}


def cosine_beta_schedule(timesteps: int, s: float = 8e-3) -> torch.Tensor:
    return CosineTables(timesteps, s).betas


def compute_alphas(betas: torch.Tensor):
    return CosineTables.from_betas(betas).as_dict()


def modulo_with_wrapped_range(vals, range_min: float = -np.pi, range_max: float = np.pi):
    """Map onto [range_min, range_max) by a floored modulo of the shifted value
    (reference utils.py:20-40; e.g. (3, -2, 2) -> -1)."""
    if not (range_min <= 0.0 and range_min < range_max):
        raise AssertionError("need range_min <= 0 < range_max")
    span = range_max - range_min
    return (vals - range_min) % span + range_min


def _wrapped_delta(prediction, truth):
    return modulo_with_wrapped_range(truth - prediction, -torch.pi, torch.pi)


def _radian_l1_elem(input, target):
    delta = (target % TWO_PI) - (input % TWO_PI)
    delta = (delta + torch.pi) % TWO_PI - torch.pi
    return delta.abs()


def radian_l1_loss(input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Mean absolute angular difference (reference utils.py:61-76); both arguments are first
    reduced mod 2*pi, exactly as the reference does."""
    return _radian_l1_elem(input, target).mean()


def radian_smooth_l1_loss(input: torch.Tensor, target: torch.Tensor, beta: float = 1.0,
                          circle_penalty: float = 0.0) -> torch.Tensor:
    """Huber-style loss on the wrapped difference (reference utils.py:78-109): quadratic inside
    ``beta``, linear outside; ``circle_penalty`` charges whole turns of ``input``."""
    return _radian_smooth_l1_elem(input, target, beta, circle_penalty).mean()


def _radian_smooth_l1_elem(input, target, beta=1.0, circle_penalty=0.0):
    if target.shape != input.shape:
        raise AssertionError(f"Mismatched shapes: {input.shape} != {target.shape}")
    if not beta > 0:
        raise AssertionError("beta must be positive")
    delta = _wrapped_delta(input, target)
    mag = delta.abs()
    loss = torch.where(mag < beta, 0.5 * (delta ** 2) / beta, mag - 0.5 * beta)
    if circle_penalty > 0:
        loss = loss + circle_penalty * torch.div(input.abs(), torch.pi, rounding_mode="trunc")
    return loss


def elementwise_form(fn):
    """The per-element form ``e`` of one of this module's loss functions (``fn(x, y) == e(x, y).mean()``), or None for a
    callable this module does not know: lets the training step take the mean over the un-padded positions as a masked
    sum -- no ``torch.where(mask)`` index lists, hence no device-to-host synchronisation inside the step."""
    import functools
    if fn is radian_l1_loss:
        return _radian_l1_elem
    if fn is radian_smooth_l1_loss:
        return _radian_smooth_l1_elem
    if isinstance(fn, functools.partial) and fn.func is radian_smooth_l1_loss and not fn.args:
        return functools.partial(_radian_smooth_l1_elem, **fn.keywords)
    return None


def tolerant_comparison_check(values, cmp, v):
    """``values`` all >= v (or <= v) up to 1e-5 (reference utils.py:111-130, unused there)."""
    if cmp not in (">=", "<="):
        raise ValueError(f"Illegal comparator: {cmp}")
    extreme = np.nanmin(values) if cmp == ">=" else np.nanmax(values)
    gap = extreme - v
    if np.isclose(gap, 0, atol=1e-5):
        return True
    return bool(gap > 0) if cmp == ">=" else bool(gap < 0)
