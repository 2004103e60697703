"""Training-time noise injection: where ``L.z`` is hot in the reference's training loop.

``noise_injection`` reproduces, on the HIP path, what one optimiser step computes before the network is called
(iadb_bn.py:870-956; latent_iadb_bn_diffusers.py:603-621):

    x0, noise_bn, noise_wn = get_noise_v2(device, x1, L, gamma_t, t, noise_type, 'train', inplace=False)
    x_alpha = alpha * x0 + (1 - alpha) * x1                       # x1 is the data, x0 the noise
    tar1 = x1 - x0;  tar2 = alpha_{t-1} * (noise_bn - noise_wn)   # out_channel == 2C
    tar  = tar1 + tar2                                            # out_channel == C

The blue-noise transform is ``bndm_bluenoise`` (same kernel as the sampler's), the blend and the targets one
elementwise kernel (``bndm_iadb_train_targets``) in the reference's fp32 operation order.  The optimiser, the
backward pass and the data pipeline are out of scope.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .bluenoise import get_noise_v2


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def train_targets(x0, x1, noise_bn, noise_wn, alpha, alpha_prev, want_x_alpha=True):
    """Blend + regression targets for tensors already on the GPU.  Returns (x_alpha, tar1, tar2, tar);
    tar2 is None when noise_bn / noise_wn are None ('gaussian' / 'GBN': tar == tar1)."""
    _lib.require_gpu(x0, "train_targets(x0)")
    _lib.require_gpu(x1, "train_targets(x1)")
    if x0.shape != x1.shape:
        raise ValueError(f"x0 {tuple(x0.shape)} and x1 {tuple(x1.shape)} differ")
    B = x0.shape[0]
    per = x0[0].numel() if B else 0
    f = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
    x0, x1, noise_bn, noise_wn = f(x0), f(x1), f(noise_bn), f(noise_wn)
    alpha = f(alpha).reshape(-1)
    alpha_prev = f(alpha_prev).reshape(-1) if alpha_prev is not None else None
    if alpha.numel() != B or (alpha_prev is not None and alpha_prev.numel() != B):
        raise ValueError("alpha / alpha_prev must have one entry per sample")
    x_alpha = torch.empty_like(x0) if want_x_alpha else None
    tar1 = torch.empty_like(x0)
    tar2 = torch.empty_like(x0) if noise_bn is not None else None
    tar = torch.empty_like(x0)
    rc = _lib.load().bndm_iadb_train_targets(_ptr(x0), _ptr(x1), _ptr(noise_bn), _ptr(noise_wn), _ptr(alpha),
                                             _ptr(alpha_prev), _ptr(x_alpha), _ptr(tar1), _ptr(tar2), _ptr(tar), B,
                                             per, _lib.current_stream_ptr())
    _lib.check(rc, "bndm_iadb_train_targets")
    return x_alpha, tar1, tar2, tar


def noise_injection(device, x1, cov_mat_L, gamma_t, alpha, alpha_prev, noise_type="gaussian", time_step=None):
    """One training step's noise injection.  ``x1`` data [B,C,H,W]; ``gamma_t``, ``alpha``, ``alpha_prev`` [B].
    Returns a dict with x0, noise_bn, noise_wn (as get_noise_v2) and x_alpha, tar1, tar2, tar."""
    x0, noise_bn, noise_wn = get_noise_v2(device, x1, cov_mat_L, gamma_t, time_step, noise_type=noise_type,
                                          train_or_test="train", inplace=False)
    two = noise_type in ("gaussianBN", "gaussianRN")            # the only types with a second target (iadb_bn.py:941)
    x_alpha, tar1, tar2, tar = train_targets(x0, x1, noise_bn if two else None, noise_wn if two else None, alpha,
                                             alpha_prev if two else None)
    return {"x0": x0, "noise_bn": noise_bn, "noise_wn": noise_wn, "x_alpha": x_alpha, "tar1": tar1, "tar2": tar2,
            "tar": tar}
