"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the tiled blue-noise generator.

numpy restatement of ``bluenoise/get_noise_recent.py`` (reference file, /root/reference).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product (``bndm_amd``) never does.

Parity pin: ``tests/golden/noise_*.npz`` were produced by importing the reference's
``get_noise_v2`` in the build container (``tests/golden/make_golden.py``); ``tests/test_oracle_noise.py``
checks this restatement against every one of them.

The restatement is written as explicit index maps rather than as the reference's
view/permute/cat chain, so that it documents *what* lands *where*:

* 64 px   (get_noise_recent.py:103-121): bn[b,c,i] = sum_j L[i,j] z[b,c,j]
* 32 px   (get_noise_recent.py:77-99):   z is the 2x2 periodic replication of x; crop [0:32,0:32]
* 128 px  (get_noise_recent.py:126-164 + noise_padding :7-19):
    columns are tile-major on the way in (f = k*B + b, k = TL,TR,BL,BR) and are re-read
    batch-major on the way out (f = 4*b' + k'); slot k' lands at rows (k'&1)*64, cols (k'>>1)*64.
    noise_wn re-interprets the [4096, C] buffer as [C, 64, 64] without un-permuting.
"""
from __future__ import annotations

import numpy as np

TILE = 64
NPIX = TILE * TILE

_TILE_ORIGIN = ((0, 0), (0, 64), (64, 0), (64, 64))      # extraction order t1..t4 (:131, :52)
_SLOT_ORIGIN = ((0, 0), (64, 0), (0, 64), (64, 64))      # noise_padding placement (:10-14)


def apply_factor(L: np.ndarray, zcols: np.ndarray) -> np.ndarray:
    """zcols [F, 4096] -> [F, 4096]; out[f, i] = sum_j L[i, j] zcols[f, j]  (:88,:113,:146).

    Dense product, exactly what torch.matmul(cov_mat_L, noise) evaluates (no use of
    triangularity), accumulated in float32 by BLAS.
    """
    return np.ascontiguousarray((L.astype(np.float32) @ zcols.astype(np.float32).T).T)


def noise_padding(slots: np.ndarray) -> np.ndarray:
    """[B,4,C,64,64] -> [B,C,128,128] (get_noise_recent.py:7-19)."""
    B, four, C = slots.shape[:3]
    assert four == 4
    out = np.empty((B, C, 128, 128), dtype=slots.dtype)
    for k, (r0, c0) in enumerate(_SLOT_ORIGIN):
        out[:, :, r0:r0 + 64, c0:c0 + 64] = slots[:, k]
    return out


def _tile_major_columns(x: np.ndarray) -> np.ndarray:
    """x [B,C,128,128] -> z [4B, C, 4096] with f = k*B + b  (:131-133)."""
    B, C = x.shape[:2]
    z = np.empty((4 * B, C, NPIX), dtype=np.float32)
    for k, (r0, c0) in enumerate(_TILE_ORIGIN):
        z[k * B:(k + 1) * B] = x[:, :, r0:r0 + 64, c0:c0 + 64].reshape(B, C, NPIX)
    return z


def _scrambled_white(z: np.ndarray, B: int) -> np.ndarray:
    """z [4B,C,4096] -> [B,4,C,64,64]: memory [f][j][c] re-read as [b'][k'][c'][y][x] (:143-144)."""
    F, C, _ = z.shape
    flat = np.ascontiguousarray(z.transpose(0, 2, 1)).reshape(-1)     # [f][j][c]
    return flat.reshape(B, 4, C, TILE, TILE)


def _blend(bn, wn, alpha_t, noise_type):
    if noise_type in ("gaussianBN", "gaussianRN"):
        a = np.asarray(alpha_t, dtype=np.float32).reshape(-1, 1, 1, 1)
        return bn * (np.float32(1) - a) + wn * a                     # (:91,:116,:160)
    if noise_type == "GBN":
        return bn                                                    # (:93,:118,:162)
    raise NotImplementedError(noise_type)


def get_noise_v2(x, L, alpha_t, noise_type="gaussian", train_or_test="train", z=None):
    """Oracle for get_noise_v2 (get_noise_recent.py:23-196).

    ``z`` replaces the RNG draw: ``None`` means ``inplace=True`` (z = x); otherwise it is the
    white sample the non-inplace branch would have drawn -- shape of ``x`` for 32/64 px and for
    'gaussian', shape [4B, C, 64, 64] for the 128-px blue-noise branch (:138).
    Returns (noise, noise_bn, noise_wn) as float32 arrays.
    """
    x = np.asarray(x, dtype=np.float32)
    B, C, res = x.shape[0], x.shape[1], x.shape[-1]

    if noise_type == "gaussian":
        if res == 64:
            noise = x if z is None else np.asarray(z, np.float32)
        elif res == 128:
            noise = x if z is None else np.asarray(z, np.float32)
            if train_or_test == "test":                              # (:50-56) rebuilt from x, not z
                noise = noise_padding(_scrambled_white(_tile_major_columns(x), B))
        else:
            raise NotImplementedError(res)                           # (:58-59)
        return noise, noise, noise

    if noise_type not in ("gaussianBN", "gaussianRN", "GBN"):
        raise NotImplementedError(noise_type)

    if res == 32:                                                    # (:77-99)
        src = x if z is None else np.asarray(z, np.float32)
        if z is None:
            src = np.tile(src, (1, 1, 2, 2))                         # cat along H then W
        # non-inplace: randn_like of the already-tiled x -> z has shape [B,C,64,64]
        wn = src.reshape(B, C, TILE, TILE)
        bn = apply_factor(L, wn.reshape(B * C, NPIX)).reshape(B, C, TILE, TILE)
        noise = _blend(bn, wn, alpha_t, noise_type)
        return noise[:, :, :32, :32], bn[:, :, :32, :32], wn[:, :, :32, :32]

    if res == 64:                                                    # (:103-121)
        wn = (x if z is None else np.asarray(z, np.float32)).reshape(B, C, TILE, TILE)
        bn = apply_factor(L, wn.reshape(B * C, NPIX)).reshape(B, C, TILE, TILE)
        return _blend(bn, wn, alpha_t, noise_type), bn, wn

    if res == 128:                                                   # (:126-164)
        zc = _tile_major_columns(x) if z is None else np.asarray(z, np.float32).reshape(4 * B, C, NPIX)
        wn = noise_padding(_scrambled_white(zc, B))
        bn_cols = apply_factor(L, zc.reshape(4 * B * C, NPIX)).reshape(4 * B, C, NPIX)
        bn = noise_padding(bn_cols.reshape(B, 4, C, TILE, TILE))     # batch-major re-read (:146)
        return _blend(bn, wn, alpha_t, noise_type), bn, wn

    raise NotImplementedError(res)                                   # (:166-167)
