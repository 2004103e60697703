"""Deterministic synthetic stand-ins for artefacts the reference downloads.

The reference loads its 4096x4096 factor from
``bluenoise/cov_gaussian{BN,RN}_L_res64_d3.npz`` (key ``'x'``; iadb_bn.py:83-86) and its
UNet weights from ``results_gaussianBN/.../model.ckpt`` (iadb_bn.py:714).  Neither file ships
with the repository (README.md:33-36) and there is no network here, so benchmarks, the CLI
and the tests synthesise them:

* ``blue_noise_factor``  -- Cholesky factor of a blue (high-pass) circulant covariance on the
  64x64 torus; same container as the reference (dense row-major, zeros above the diagonal).
* ``formula_factor``     -- a lower-triangular matrix given by exact integer arithmetic, so
  golden vectors do not depend on the LAPACK build.
"""
from __future__ import annotations

import os
import numpy as np

TILE = 64
N = TILE * TILE  # 4096


def blue_noise_factor(kind: str = "blue", sigma: float = 0.18, floor: float = 1e-3) -> np.ndarray:
    """L with L L^T = Sigma, Sigma circulant with spectrum P(f)=1-exp(-|f|^2/2s^2)+floor.

    kind='blue' for the BN file, 'red' for the RN file (P = exp(-|f|^2/2s^2)+floor).
    Returns float32 [4096,4096], exact zeros above the diagonal, unit-variance rows.
    """
    f = np.fft.fftfreq(TILE)
    fy, fx = np.meshgrid(f, f, indexing="ij")
    r2 = fx * fx + fy * fy
    g = np.exp(-r2 / (2.0 * sigma * sigma))
    spec = (1.0 - g if kind == "blue" else g) + floor
    acf = np.real(np.fft.ifft2(spec))          # autocorrelation on the torus
    acf /= acf[0, 0]
    iy, ix = np.divmod(np.arange(N), TILE)
    # Sigma[p, q] = acf[(py-qy) mod 64, (px-qx) mod 64], built row-block-wise to bound memory
    sigma_m = np.empty((N, N), dtype=np.float64)
    for p0 in range(0, N, 512):
        dy = (iy[p0:p0 + 512, None] - iy[None, :]) % TILE
        dx = (ix[p0:p0 + 512, None] - ix[None, :]) % TILE
        sigma_m[p0:p0 + 512] = acf[dy, dx]
    chol = np.linalg.cholesky(sigma_m)
    return np.ascontiguousarray(np.tril(chol).astype(np.float32))


def formula_factor() -> np.ndarray:
    """Exactly reproducible lower-triangular float32 matrix (integer arithmetic / 2^k).

    L[i,j] = ((131*i + 71*j + (i*j) % 97) % 257 - 128) / 8192 for j<i, L[i,i] = 1, 0 above.
    """
    i = np.arange(N, dtype=np.int64)[:, None]
    j = np.arange(N, dtype=np.int64)[None, :]
    v = ((131 * i + 71 * j + (i * j) % 97) % 257 - 128).astype(np.float32) / np.float32(8192.0)
    v = np.tril(v, -1)
    v[np.arange(N), np.arange(N)] = 1.0
    return np.ascontiguousarray(v)


def load_or_make_factor(path: str, kind: str = "blue") -> np.ndarray:
    """np.load(path)['x'] as the reference does (iadb_bn.py:83); synthesise when absent."""
    if os.path.exists(path):
        return np.load(path)["x"].astype(np.float32)
    return blue_noise_factor(kind)
