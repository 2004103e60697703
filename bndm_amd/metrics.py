"""Image-quality metrics printed by the conditional sampler (iadb_bn.py:570,636-644,681: ``piq.psnr`` / ``piq.ssim``).

``piq`` is an un-vendored PyPI dependency of the reference that is not installed here; these functions restate its
published defaults (SSIM: 11x11 Gaussian window, sigma 1.5, k1 = 0.01, k2 = 0.03, "valid" convolution, per-channel
maps averaged over space and channels, inputs average-pooled by max(1, round(min(H, W) / 256)) first).  They are
reporting code on tensors the sampler already produced -- not part of the hot path -- and unpinned against piq.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def psnr(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """Per-image PSNR in dB of [B, C, H, W] tensors in [0, data_range]."""
    mse = torch.mean((x.double() - y.double()) ** 2, dim=(1, 2, 3)).clamp_min(1e-12)
    return 10.0 * torch.log10(data_range ** 2 / mse)


def _gaussian_kernel(size: int, sigma: float, dtype, device) -> torch.Tensor:
    c = torch.arange(size, dtype=dtype, device=device) - (size - 1) / 2.0
    g = torch.exp(-(c ** 2) / (2.0 * sigma ** 2))
    k = g[:, None] * g[None, :]
    return k / k.sum()


def ssim(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0, kernel_size: int = 11, kernel_sigma: float = 1.5,
         k1: float = 0.01, k2: float = 0.03) -> torch.Tensor:
    """Per-image SSIM (reduction='none') of [B, C, H, W] tensors in [0, data_range]."""
    if x.shape != y.shape or x.dim() != 4:
        raise ValueError(f"ssim: shapes {tuple(x.shape)} / {tuple(y.shape)}")
    x = x.double() / data_range
    y = y.double() / data_range
    f = max(1, round(min(x.shape[-2:]) / 256))
    if f > 1:
        x, y = F.avg_pool2d(x, f), F.avg_pool2d(y, f)
    C = x.shape[1]
    k = _gaussian_kernel(kernel_size, kernel_sigma, x.dtype, x.device).expand(C, 1, kernel_size, kernel_size)
    c1, c2 = k1 ** 2, k2 ** 2
    mu_x, mu_y = F.conv2d(x, k, groups=C), F.conv2d(y, k, groups=C)
    mu_xx, mu_yy, mu_xy = mu_x * mu_x, mu_y * mu_y, mu_x * mu_y
    s_xx = F.conv2d(x * x, k, groups=C) - mu_xx
    s_yy = F.conv2d(y * y, k, groups=C) - mu_yy
    s_xy = F.conv2d(x * y, k, groups=C) - mu_xy
    cs = (2.0 * s_xy + c2) / (s_xx + s_yy + c2)
    ss = (2.0 * mu_xy + c1) / (mu_xx + mu_yy + c1) * cs
    return ss.mean(dim=(-1, -2)).mean(dim=1)
