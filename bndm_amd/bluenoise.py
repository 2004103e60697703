"""Host-side mirror of the reference's ``bluenoise/get_noise_recent.py`` on top of the HIP kernels.

``get_noise_v2`` keeps the reference signature and return triple
(get_noise_recent.py:23,196); the arithmetic (L.z over 64x64 tiles, tile gather, batch-major
re-read, noise_padding, white<->blue lerp) runs in ``bndm_bluenoise`` (csrc/bluenoise.hip).
Only the RNG draws of the non-inplace branches stay in PyTorch (``torch.randn`` on the tensor's
device), with the reference's draw shapes (:108, :138) so the RNG contract is unchanged.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

Z_COLUMNS, Z_IMAGE32, Z_IMAGE128 = 0, 1, 2
MODE_BLEND, MODE_PURE_BN, MODE_SCRAMBLE = 0, 1, 2

_BN_TYPES = ("gaussianBN", "gaussianRN", "GBN")


def _is_lower_triangular(L: torch.Tensor) -> bool:
    """True when L is exactly zero above the diagonal (then only j<=i is read).  Probed once per tensor object
    and version -- the result rides on the tensor itself, so a new factor allocated at a recycled address is
    probed again; a factor with anything above the diagonal takes the dense path, which is the reference's
    semantics for arbitrary matrices.  The probe reads the 64 MiB matrix and synchronises (one ``.item()``):
    pass ``l_is_triangular=`` to ``get_noise_v2`` to skip it."""
    hit = getattr(L, "_bndm_tri", None)
    if hit is None or hit[0] != L._version:
        hit = (L._version, bool((torch.triu(L, diagonal=1) == 0).all().item()))
        L._bndm_tri = hit
    return hit[1]


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _run(L, z, z_layout, alpha, B_global, b_begin, b_count, Cch, res, mode, want_parts=True, tri=None):
    lib = _lib.load()
    dev = z.device
    shape = (b_count, Cch, res, res)
    noise = torch.empty(shape, dtype=torch.float32, device=dev)
    if mode == MODE_SCRAMBLE:
        noise_bn = noise_wn = None
        ws, ws_bytes, dense = None, 0, 0
    else:
        noise_bn = torch.empty(shape, dtype=torch.float32, device=dev) if want_parts else None
        noise_wn = torch.empty(shape, dtype=torch.float32, device=dev) if want_parts else None
        ws_bytes = lib.bndm_bluenoise_workspace_bytes(b_count, Cch, res)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        dense = 0 if (_is_lower_triangular(L) if tri is None else tri) else 1
    rc = lib.bndm_bluenoise(_ptr(L), dense, _ptr(z), z_layout, _ptr(alpha), _ptr(noise), _ptr(noise_bn),
                            _ptr(noise_wn), B_global, b_begin, b_count, Cch, res, mode, _ptr(ws), ws_bytes,
                            _lib.current_stream_ptr())
    _lib.check(rc, "bndm_bluenoise")
    return noise, noise_bn, noise_wn


def _check_inputs(x, cov_mat_L, alpha_t, need_L):
    _lib.require_gpu(x, "get_noise_v2(x)")
    if x.dtype != torch.float32:
        raise TypeError(f"get_noise_v2: x must be float32, got {x.dtype}")
    if need_L:
        _lib.require_gpu(cov_mat_L, "get_noise_v2(cov_mat_L)")
        if cov_mat_L.dtype != torch.float32 or tuple(cov_mat_L.shape) != (4096, 4096):
            raise ValueError("get_noise_v2: cov_mat_L must be float32 [4096, 4096]")
        if not cov_mat_L.is_contiguous():
            raise ValueError("get_noise_v2: cov_mat_L must be contiguous")


def noise_padding(noise_small, res=128):
    """[B,4,C,64,64] -> [B,C,128,128]; slot k lands at rows (k&1)*64, cols (k>>1)*64
    (get_noise_recent.py:7-19).  Pure data movement, kept for API parity."""
    if res != 128:
        raise NotImplementedError
    B, _, Cc = noise_small.shape[:3]
    out = noise_small.new_empty((B, Cc, 128, 128))
    for k in range(4):
        r0, c0 = (k & 1) * 64, (k >> 1) * 64
        out[:, :, r0:r0 + 64, c0:c0 + 64] = noise_small[:, k]
    return out


def get_noise_v2(device, x, cov_mat_L, alpha_t, time_step, noise_type='gaussian', train_or_test='train',
                 inplace=False, *, batch_range=None, global_z=None, l_is_triangular=None):
    """Drop-in for bluenoise.get_noise_recent.get_noise_v2 (get_noise_recent.py:23).

    Extra keyword-only arguments (not in the reference) serve batch sharding across GPUs
    (SURVEY.md 8e): ``batch_range=(b_begin, b_count)`` computes only those samples of the global
    batch ``x`` -- the 128-px branch mixes tiles across the batch, so every rank passes the global
    ``x`` (or ``global_z``) and its own range.  ``l_is_triangular`` (True / False) states whether ``cov_mat_L`` is
    exactly zero above its diagonal (a Cholesky factor is); ``None`` probes the matrix once per tensor, which
    reads 64 MiB and synchronises on first use.
    """
    res = x.shape[-1]
    Cch = x.shape[1]
    B = x.shape[0]
    b_begin, b_count = (0, B) if batch_range is None else batch_range

    if noise_type == 'gaussian':
        if res == 64:
            noise = x if inplace else torch.randn_like(x)
        elif res == 128:
            noise = x if inplace else torch.randn_like(x)
            if train_or_test == 'test':
                # rebuilt from x, never from the fresh draw (:51-56)
                _check_inputs(x, None, None, need_L=False)
                noise, _, _ = _run(None, x.contiguous(), Z_IMAGE128, None, B, b_begin, b_count, Cch, 128,
                                   MODE_SCRAMBLE)
        else:
            raise NotImplementedError
        if batch_range is not None and noise.shape[0] == B and b_count != B:
            noise = noise[b_begin:b_begin + b_count]
        return noise, noise, noise

    if noise_type == 'uniform':
        # the reference draws rand_like and then fails at its return statement because noise_bn
        # was never bound (:69-71,:196); keep the failure, skip the wasted draw.
        raise UnboundLocalError("local variable 'noise_bn' referenced before assignment")

    if noise_type not in _BN_TYPES:
        raise NotImplementedError

    if res not in (32, 64, 128):
        raise NotImplementedError
    _check_inputs(x, cov_mat_L, alpha_t, need_L=True)
    mode = MODE_PURE_BN if noise_type == 'GBN' else MODE_BLEND
    alpha = None
    if mode == MODE_BLEND:
        alpha = alpha_t.reshape(-1).to(device=x.device, dtype=torch.float32).contiguous()
        if alpha.numel() != B:
            raise RuntimeError(f"alpha_t has {alpha.numel()} entries for a batch of {B}")

    if res == 32:
        if inplace:
            z, layout = x.contiguous(), Z_IMAGE32
        else:
            z = global_z if global_z is not None else torch.randn((B, Cch, 64, 64), dtype=x.dtype, device=x.device)
            layout = Z_COLUMNS
    elif res == 64:
        if inplace:
            z = x.view(B, Cch, -1).view(B, Cch, 64, 64)      # same contiguity requirement as :111
        else:
            z = global_z if global_z is not None else torch.randn_like(x)
        z, layout = z.contiguous(), Z_COLUMNS
    else:
        if inplace:
            z, layout = x.contiguous(), Z_IMAGE128
        else:
            # the reference draws this one from the CPU generator and copies it over (:138);
            # same here so a seeded run consumes the same stream
            z = global_z if global_z is not None else torch.randn(B * 4, Cch, 64, 64).float().to(device)
            z, layout = z.contiguous(), Z_COLUMNS
    return _run(cov_mat_L, z, layout, alpha, B, b_begin, b_count, Cch, res, mode, tri=l_is_triangular)


# README.md:33 calls it ``get_noise``; the code never defined that name.  Provide the alias.
get_noise = get_noise_v2
