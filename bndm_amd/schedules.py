"""alpha / gamma schedules of the reference, plus the host-side step tables the HIP loop consumes.

``get_scheduler`` / ``get_scheduler_gamma`` keep the signatures of utils.py:94,120 (the library
form of iadb_bn.py:90,147).  The sampling loop never evaluates them per step on the device as the
reference does (~60 tiny launches + one H2D copy per step): ``step_tables`` evaluates
them once on the host, in float32 with the reference's operation order, for t = 0..N.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def _squash(kind, v, tau):
    if kind == "sigmoid":
        return torch.sigmoid(v / tau)
    if kind == "cosine":
        return torch.pow(torch.cos(v * np.pi / 2.0), 2.0 * tau)
    raise NotImplementedError(kind)


def get_scheduler(x, scheduler, nb_steps, scheduler_param=0.02):
    """alpha(t).  'linear' is the only mode of utils.py:94-116 and of every shipped script; the
    'sigmoid' / 'cosine' variants of iadb_bn.py:109-138 (fixed end points) are kept for the CLI."""
    kind = scheduler.lower()
    if kind == "linear":
        return x / nb_steps
    if not torch.is_tensor(x):
        raise NotImplementedError("non-linear alpha schedules take tensors")
    if kind == "sigmoid":
        lo, hi, tau = float(scheduler_param), 3.0, 0.9
    elif kind == "cosine":
        lo, hi, tau = 0.2, 1.0, scheduler_param
    else:
        raise NotImplementedError
    one = torch.ones_like(x)
    f_lo, f_hi = _squash(kind, one * lo, tau), _squash(kind, one * hi, tau)
    f_t = _squash(kind, (x / nb_steps) * (one * hi - one * lo) + one * lo, tau)
    return 1 - torch.clamp((f_hi - f_t) / (f_hi - f_lo), 1e-9, 1.0)


def get_scheduler_gamma(x, scheduler, scheduler_params, nb_steps):
    """gamma(t) with scheduler_params = (tau, start, end)  (utils.py:120-174)."""
    kind = scheduler.lower()
    if kind == "linear":
        return x / nb_steps
    if kind not in ("sigmoid", "cosine"):
        raise NotImplementedError
    tau, lo, hi = scheduler_params[0], scheduler_params[1], scheduler_params[2]
    start, end = torch.ones_like(x) * lo, torch.ones_like(x) * hi
    f_lo, f_hi = _squash(kind, start, tau), _squash(kind, end, tau)
    f_t = _squash(kind, (x / nb_steps) * (end - start) + start, tau)
    return 1 - torch.clamp((f_hi - f_t) / (f_hi - f_lo), 1e-9, 1)


def step_tables(nb_step, scheduler_alpha="linear", scheduler_gamma="linear", scheduler_params=(1.0, 0.0, 3.0),
                alpha_param=0.02):
    """float32 numpy tables over s = 0..nb_step-1 (t = nb_step-1-s):
    t_in = alpha(t+1) (the model's time input, utils.py:211), da = alpha(t+1)-alpha(t),
    dg = gamma(t+1)-gamma(t) (utils.py:218-221)."""
    t = torch.arange(nb_step - 1, -1, -1, dtype=torch.int64)
    params = scheduler_params
    if torch.is_tensor(params):
        params = params.detach().cpu().float()
    else:
        params = torch.tensor([float(p) for p in params], dtype=torch.float32)
    a1 = get_scheduler((t + 1).float(), scheduler_alpha, nb_step, alpha_param)
    a0 = get_scheduler(t.float(), scheduler_alpha, nb_step, alpha_param)
    g1 = get_scheduler_gamma((t + 1).float(), scheduler_gamma, params, nb_step)
    g0 = get_scheduler_gamma(t.float(), scheduler_gamma, params, nb_step)
    f = lambda v: np.ascontiguousarray(v.numpy().astype(np.float32))
    return f(a1), f(a1 - a0), f(g1 - g0)


def ddim_tables(num_inference_steps, num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02,
                beta_schedule="linear"):
    """DDIMScheduler(num_train_timesteps, beta_schedule) defaults (ddim_diffusers.py:499-503):
    returns (timesteps int64 descending, coef float32 [n,5] = t, sqrt(a_t), sqrt(1-a_t), sqrt(a_prev),
    sqrt(1-a_prev))."""
    if beta_schedule == "linear":
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    elif beta_schedule == "scaled_linear":
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    else:
        raise NotImplementedError(beta_schedule)
    acp = torch.cumprod(1.0 - betas, dim=0)
    ratio = num_train_timesteps // num_inference_steps
    ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
    coef = np.zeros((num_inference_steps, 5), dtype=np.float32)
    for i, t in enumerate(ts):
        a_t = acp[t]
        a_p = acp[t - ratio] if t - ratio >= 0 else torch.tensor(1.0)
        coef[i] = [float(t), float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_p ** 0.5), float((1 - a_p) ** 0.5)]
    return ts, coef, acp
