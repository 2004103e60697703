"""IADB / DDIM sampling loops on the HIP engine, behind the reference's function signatures.

* ``sample_iadb`` -- utils.py:179-240 (library form of iadb_bn.py:286-379).  When ``model`` is the
  HIP ``UNet2DModel`` the whole loop runs inside ``bndm_unet_sample_iadb`` (one C call: schedule
  tables precomputed on the host, Euler update fused into one kernel per step, no per-step host
  maths or H2D copies).  Any other callable is driven step by step with the HIP Euler kernel.
* ``sample_iadb_conditional`` -- iadb_bn.py:384-438 (x_c concatenated on channels each step).
* ``get_model`` -- utils.py:7-84 / iadb_bn.py:205-282.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np
import torch

from . import _lib
from .schedules import get_scheduler, get_scheduler_gamma, step_tables
from .unet import UNet2DModel, unwrap

_LEVELS = {
    64: ((128, 128, 256, 256, 512, 512), 4),
    128: ((128, 128, 128, 256, 256, 512, 512), 5),
    256: ((128, 128, 128, 128, 256, 256, 512, 512), 6),
}


def get_model(inp_channel=3, out_channel=3, res=64, activation="silu", **kw):
    """utils.get_model (utils.py:7): UNet with the reference's per-resolution block layout;
    attention sits in the second-to-last down block and the second up block."""
    if res not in _LEVELS:
        raise NotImplementedError
    boc, attn_down = _LEVELS[res]
    n = len(boc)
    down = tuple("AttnDownBlock2D" if i == attn_down else "DownBlock2D" for i in range(n))
    up = tuple("AttnUpBlock2D" if i == 1 else "UpBlock2D" for i in range(n))
    return UNet2DModel(block_out_channels=boc, out_channels=out_channel, in_channels=inp_channel,
                       up_block_types=up, down_block_types=down, act_fn=activation, add_attention=True, **kw)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _snap_mask(nb_step, train_or_test, log_freq):
    mask = np.zeros(nb_step, dtype=np.uint8)
    if train_or_test == "test":
        lf = 100 if nb_step == 1000 else log_freq
        for s in range(nb_step):
            t = nb_step - 1 - s
            if t % lf == 0 or t == nb_step - 1:
                mask[s] = 1
    return mask


def _iadb_loop(model, x0, x_c, nb_step, scheduler_alpha, scheduler_gamma, scheduler_params, out_channel,
               noise_type, train_or_test, log_freq, alpha_param=0.02, tables=None):
    _lib.require_gpu(x0, "sample_iadb(x0)")
    if noise_type not in ("gaussianBN", "gaussianRN", "gaussian", "GBN"):
        raise NotImplementedError
    lib = _lib.load()
    B, Cc = x0.shape[0], x0.shape[1]
    if noise_type in ("gaussianBN", "gaussianRN"):
        if out_channel not in (Cc, 2 * Cc):
            raise NotImplementedError
        use_gamma = out_channel == 2 * Cc
    else:
        use_gamma = False
    t_in, da, dg = tables if tables is not None else \
        step_tables(nb_step, scheduler_alpha, scheduler_gamma, scheduler_params, alpha_param)
    mask = _snap_mask(nb_step, train_or_test, log_freq)
    x = x0.detach().to(torch.float32).contiguous().clone()
    core = unwrap(model)
    n_snap = int(mask.sum())
    snaps = torch.empty((n_snap,) + tuple(x.shape), dtype=torch.float32, device=x.device) if n_snap else None
    times = []

    if isinstance(core, UNet2DModel):
        Cout = core.config["out_channels"]
        if not use_gamma and Cout != Cc:
            # reference: x + da * d broadcasts only when d has C channels
            raise RuntimeError(f"model emits {Cout} channels but noise_type={noise_type!r} uses {Cc}")
        h = core._ensure_engine(B, x.shape[-1], x.device)
        xc = x_c.detach().to(torch.float32).contiguous() if x_c is not None else None
        dgz = dg if use_gamma else np.zeros_like(dg)
        # mean_forward_time: the reference brackets model(...) with time.time() WITHOUT a device synchronise
        # (iadb_bn.py:318-321), i.e. it reports launch time.  Here the whole loop is one asynchronous C call, so the
        # figure is taken from device events around it: (loop time) / steps, forward + Euler update, in seconds.
        # The events go onto x's device and onto the very stream handed to the library.
        with torch.cuda.device(x.device):
            st = torch.cuda.current_stream(x.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = lib.bndm_unet_sample_iadb(h, _ptr(x), _ptr(xc), B, Cc, nb_step, t_in.ctypes.data_as(C.c_void_p),
                                           da.ctypes.data_as(C.c_void_p), dgz.ctypes.data_as(C.c_void_p),
                                           mask.ctypes.data_as(C.c_void_p), _ptr(snaps), C.c_void_p(st.cuda_stream))
            _lib.check(rc, "bndm_unet_sample_iadb")
            e1.record(st)
        if train_or_test == "test":                     # only the 'test' form returns the figure (utils.py:237-240)
            e1.synchronize()                            # (the reference's 'test' callers move x to the host right after)
            times = [e0.elapsed_time(e1) * 1e-3 / max(nb_step, 1)] * 2
    else:
        k = 0
        for s in range(nb_step):
            tt = torch.full((B,), float(t_in[s]), dtype=torch.float32, device=x.device)
            inp = x if x_c is None else torch.cat([x, x_c], 1)
            t0 = time.time()
            d = model(inp, tt, return_dict=False)[0]
            times.append(time.time() - t0)
            d = d.to(torch.float32).contiguous()
            Cout = d.shape[1]
            if not use_gamma and Cout != Cc:
                raise RuntimeError(f"model emits {Cout} channels but noise_type={noise_type!r} uses {Cc}")
            rc = lib.bndm_iadb_step(_ptr(x), _ptr(d), float(da[s]), float(dg[s]) if use_gamma else 0.0, B, Cc,
                                    Cout if use_gamma else Cc, x.shape[2] * x.shape[3], _lib.current_stream_ptr()) \
                if (use_gamma or Cout == Cc) else -1
            _lib.check(rc, "bndm_iadb_step")
            if mask[s]:
                snaps[k].copy_(x)
                k += 1
    x_all = [snaps[i] for i in range(n_snap)] if n_snap else []
    return x, x_all, (float(np.mean(times[1:])) if len(times) > 1 else 0.0)


@torch.no_grad()
def sample_iadb(model, x0, nb_step, scheduler_gamma, scheduler_params, out_channel, noise_type, train_or_test,
                scheduler_alpha='linear', log_freq=1, alpha_param=0.02):
    """utils.sample_iadb (utils.py:179).  Returns (x, x_all, mean_forward_time) in 'test' mode and x
    otherwise.  ``mean_forward_time`` is NOT the reference's un-synchronised host time around model(...)
    (iadb_bn.py:318-321, i.e. launch time): with the in-engine loop it is (device time of the whole loop) / steps --
    forward + Euler update + snapshots, in seconds, from events on the stream the library ran on; reading it
    synchronises that stream.  ``log_freq`` (extra, default = utils.py's 1) selects iadb_bn.py's cadence of 25;
    ``alpha_param`` is the sigmoid start / cosine tau of a non-linear ``scheduler_alpha`` -- iadb_bn.py passes
    ``opt.scheduler_param`` there (iadb_bn.py:115,131)."""
    x, x_all, ft = _iadb_loop(model, x0, None, nb_step, scheduler_alpha, scheduler_gamma, scheduler_params,
                              out_channel, noise_type, train_or_test, log_freq, alpha_param)
    if train_or_test == 'test':
        return x, x_all, ft
    return x


@torch.no_grad()
def sample_iadb_conditional(model, x0, x_c, nb_step, scheduler_gamma, scheduler_params, out_channel, noise_type,
                            train_or_test, scheduler_alpha='linear', log_freq=25, alpha_param=0.02):
    """iadb_bn.sample_iadb_conditional (iadb_bn.py:384): model(cat([x, x_c], 1), alpha)."""
    x, x_all, _ = _iadb_loop(model, x0, x_c, nb_step, scheduler_alpha, scheduler_gamma, scheduler_params,
                             out_channel, noise_type, train_or_test, log_freq, alpha_param)
    if train_or_test == 'test':
        return x, x_all
    return x


def export_u8(x, rounding="trunc"):
    """clamp((x+1)/2,0,1)*255 -> uint8 NHWC on the device (iadb_bn.py:815-816; 'round' for
    ddim_diffusers.py:687-688)."""
    _lib.require_gpu(x, "export_u8(x)")
    x = x.detach().to(torch.float32).contiguous()
    B, Cc, H, W = x.shape
    out = torch.empty((B, H, W, Cc), dtype=torch.uint8, device=x.device)
    rc = _lib.load().bndm_export_u8(_ptr(x), _ptr(out), B, Cc, H * W, 0 if rounding == "trunc" else 1,
                                    _lib.current_stream_ptr())
    _lib.check(rc, "bndm_export_u8")
    return out
