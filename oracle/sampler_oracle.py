"""TEST INFRASTRUCTURE ONLY -- CPU oracle for schedules, the IADB/DDIM loops and image export.

torch-CPU float32 restatement of
  * ``utils.py:94-174`` / ``iadb_bn.py:90-201``   (alpha / gamma schedules)
  * ``utils.py:179-240`` / ``iadb_bn.py:286-379`` (``sample_iadb``)
  * ``iadb_bn.py:384-438``                        (``sample_iadb_conditional``)
  * ``latent_iadb_bn_diffusers.py:84-122``        (``IADBScheduler.step``)
  * ``ddim_diffusers.py:639-640,674-688``         (DDIM loop; scheduler maths restated from the
    published DDIM update, diffusers not installed -> that part is "parity unpinned")
  * ``iadb_bn.py:810-816`` / ``ddim_diffusers.py:687-688`` (uint8 export)

Pins: ``tests/golden/schedules.npz`` and ``tests/golden/loop_*.npz`` were generated from the
imported reference ``utils.py`` (see ``tests/golden/make_golden.py``).
Only tests/, smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

import math
import numpy as np
import torch


# ----------------------------------------------------------------------------- schedules
def alpha_schedule(t: torch.Tensor, nb_steps: int, kind: str = "linear", param: float = 0.02) -> torch.Tensor:
    """iadb_bn.py:90-143 (utils.py:94-116 implements 'linear' only)."""
    kind = kind.lower()
    if kind == "linear":
        return t / nb_steps
    u = t / nb_steps
    if kind == "sigmoid":                       # start = param, end = 3, tau = 0.9   (:109-123)
        s, e, tau = float(param), 3.0, 0.9
        f = lambda v: torch.sigmoid(v / tau)
        vs, ve = f(torch.full_like(t, s)), f(torch.full_like(t, e))
        out = (ve - f(u * (e - s) + s)) / (ve - vs)
    elif kind == "cosine":                      # start = .2, end = 1, exponent 2*param (:125-138)
        s, e = 0.2, 1.0
        f = lambda v: torch.cos(v * np.pi / 2) ** (2 * param)
        vs, ve = f(torch.full_like(t, s)), f(torch.full_like(t, e))
        out = (ve - f(u * (e - s) + s)) / (ve - vs)
    else:
        raise NotImplementedError(kind)
    return 1 - torch.clamp(out, 1e-9, 1.0)


def gamma_schedule(t: torch.Tensor, nb_steps: int, kind: str, params) -> torch.Tensor:
    """utils.py:120-174 == iadb_bn.py:147-201.  params = (tau, start, end)."""
    kind = kind.lower()
    if kind == "linear":
        return t / nb_steps
    tau, s, e = params[0], params[1], params[2]
    start = torch.ones_like(t) * s
    end = torch.ones_like(t) * e
    u = t / nb_steps
    if kind == "sigmoid":
        f = lambda v: torch.sigmoid(v / tau)
    elif kind == "cosine":
        f = lambda v: torch.pow(torch.cos(v * np.pi / 2.0), 2.0 * tau)
    else:
        raise NotImplementedError(kind)
    out = (f(end) - f(u * (end - start) + start)) / (f(end) - f(start))
    return 1 - torch.clamp(out, 1e-9, 1.0)


# ----------------------------------------------------------------------------- IADB loop
@torch.no_grad()
def sample_iadb(model, x0, nb_step, scheduler_gamma, scheduler_params, out_channel, noise_type,
                train_or_test, scheduler_alpha="linear", log_freq=None, x_c=None):
    """utils.py:179-240.  ``log_freq=None`` -> utils.py cadence (1, or 100 when N==1000);
    pass 25 for iadb_bn.py:368-373.  ``x_c`` turns it into sample_iadb_conditional (:384-438)."""
    x = x0
    B = x0.shape[0]
    C = x0.shape[1]
    snaps = []
    for t in range(nb_step - 1, -1, -1):
        tt = torch.full((B,), t, dtype=torch.int64)
        a1 = alpha_schedule((tt + 1).float(), nb_step, scheduler_alpha)
        a0 = alpha_schedule(tt.float(), nb_step, scheduler_alpha)
        g1 = gamma_schedule((tt + 1).float(), nb_step, scheduler_gamma, scheduler_params)
        g0 = gamma_schedule(tt.float(), nb_step, scheduler_gamma, scheduler_params)
        inp = x if x_c is None else torch.cat([x, x_c], 1)
        d = model(inp, a1, return_dict=False)[0]
        da = (a1 - a0).view(-1, 1, 1, 1)
        if noise_type in ("gaussianBN", "gaussianRN"):
            if out_channel == C:
                x = x + da * d
            elif out_channel == 2 * C:
                x = x + da * d[:, :C] + (g1 - g0).view(-1, 1, 1, 1) * d[:, C:]
            else:
                raise NotImplementedError
        elif noise_type in ("gaussian", "GBN"):
            x = x + da * d
        else:
            raise NotImplementedError
        if train_or_test == "test":
            lf = log_freq if log_freq is not None else (100 if nb_step == 1000 else 1)
            if nb_step == 1000:
                lf = 100
            if t % lf == 0 or t == nb_step - 1:
                snaps.append(x)
    if train_or_test == "test":
        return x, snaps
    return x


def iadb_scheduler_step(d, t, x, num_inference_steps, noise_type, out_channels, lat_channels=4):
    """latent_iadb_bn_diffusers.py:84-122 (alpha == gamma == linear)."""
    a, an = (t + 1) / num_inference_steps, t / num_inference_steps
    if noise_type in ("gaussianBN", "gaussianRN"):
        if out_channels == lat_channels:
            return x + (a - an) * d
        if out_channels == 2 * lat_channels:
            return x + (a - an) * d[:, :lat_channels] + (a - an) * d[:, lat_channels:]
        raise NotImplementedError
    if noise_type == "gaussian":
        return x + (a - an) * d
    raise NotImplementedError


# ----------------------------------------------------------------------------- DDIM
def ddim_tables(num_train=1000, beta_start=1e-4, beta_end=0.02, num_inference=250):
    """DDIMScheduler defaults used at ddim_diffusers.py:499-503: linear betas, 'leading' spacing,
    steps_offset 0, set_alpha_to_one=True, eta 0, clip_sample True (+-1), epsilon prediction.
    (restated from the published algorithm; diffusers is not installed -> unpinned)."""
    betas = torch.linspace(beta_start, beta_end, num_train, dtype=torch.float32)
    acp = torch.cumprod(1.0 - betas, dim=0)
    ratio = num_train // num_inference
    timesteps = (np.arange(0, num_inference) * ratio).round()[::-1].copy().astype(np.int64)
    return acp, timesteps, ratio


def ddim_step(eps, t, x, acp, ratio, clip=1.0):
    prev_t = t - ratio
    a_t = acp[t]
    a_p = acp[prev_t] if prev_t >= 0 else torch.tensor(1.0)
    x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    x0 = x0.clamp(-clip, clip)
    # use_clipped_model_output defaults to False: the direction term keeps the raw eps
    return a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps


@torch.no_grad()
def sample_ddim(model, x, num_inference=250):
    """ddim_diffusers.py:672-683."""
    acp, timesteps, ratio = ddim_tables(num_inference=num_inference)
    for t in timesteps:
        eps = model(x, torch.tensor(int(t)), return_dict=False)[0]
        x = ddim_step(eps, int(t), x, acp, ratio)
    return x


# ----------------------------------------------------------------------------- export
def export_u8(x: torch.Tensor, rounding: str = "trunc") -> np.ndarray:
    """[B,C,H,W] f32 -> [B,H,W,C] u8.  'trunc': iadb_bn.py:815-816; 'round': ddim_diffusers.py:687-688."""
    if rounding == "trunc":
        y = torch.clamp((x + 1) / 2.0, 0.0, 1.0).permute(0, 2, 3, 1).numpy() * 255
        return y.astype(np.uint8)
    y = (x / 2 + 0.5).clamp(0, 1)
    return (y.permute(0, 2, 3, 1) * 255).round().to(torch.uint8).numpy()


def train_targets(x0, x1, noise_bn, noise_wn, alpha, alpha_prev):
    """Training-time blend and regression targets (iadb_bn.py:915,946-956; latent_iadb_bn_diffusers.py:610,
    617-620): x1 is the data, x0 the noise returned by get_noise_v2(..., 'train', inplace=False).
    Returns (x_alpha, tar1, tar2, tar); tar2 is None and tar == tar1 without noise_bn / noise_wn."""
    a = alpha.view(-1, 1, 1, 1)
    x_alpha = a * x0 + (1 - alpha).view(-1, 1, 1, 1) * x1
    tar1 = x1 - x0
    if noise_bn is None:
        return x_alpha, tar1, None, tar1.clone()
    tar2 = alpha_prev.view(-1, 1, 1, 1) * (noise_bn - noise_wn)
    return x_alpha, tar1, tar2, x1 - x0 + alpha_prev.view(-1, 1, 1, 1) * (noise_bn - noise_wn)

