"""Scheduler objects with the diffusers call surface the reference scripts use.

* ``DDIMScheduler`` -- ``set_timesteps(n)``, ``.timesteps``, ``.step(eps, t, x).prev_sample``
  (ddim_diffusers.py:499-505,639-640,680); the update runs in ``bndm_ddim_step``.
* ``IADBScheduler`` -- latent_iadb_bn_diffusers.py:75-142; ``step`` runs in ``bndm_iadb_step``.
Both add ``sample(model, x)``: the whole loop inside the engine (one C call).
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from .schedules import ddim_tables
from .unet import UNet2DModel, unwrap


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


@dataclass
class SchedulerOutput:
    prev_sample: torch.Tensor


class _ConfigIO:
    _config_name = "scheduler_config.json"

    def save_pretrained(self, directory):
        os.makedirs(directory, exist_ok=True)
        with open(os.path.join(directory, self._config_name), "w") as f:
            json.dump(dict(self.config, _class_name=type(self).__name__), f, indent=2)

    @classmethod
    def from_pretrained(cls, directory, **kw):
        with open(os.path.join(directory, cls._config_name)) as f:
            cfg = json.load(f)
        cfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
        cfg.update(kw)
        accepted = cls.__init__.__code__.co_varnames
        return cls(**{k: v for k, v in cfg.items() if k in accepted})


class DDIMScheduler(_ConfigIO):
    def __init__(self, num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, clip_sample_range=1.0, prediction_type="epsilon"):
        if prediction_type != "epsilon":
            raise NotImplementedError(prediction_type)
        self.config = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule=beta_schedule, clip_sample=clip_sample,
                           clip_sample_range=clip_sample_range, prediction_type=prediction_type)
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)
        self._coef = None

    def set_timesteps(self, num_inference_steps, device=None):
        c = self.config
        ts, coef, _ = ddim_tables(num_inference_steps, c["num_train_timesteps"], c["beta_start"], c["beta_end"],
                                  c["beta_schedule"])
        self.num_inference_steps = num_inference_steps
        self.timesteps = torch.from_numpy(ts)
        self._coef = {int(t): coef[i] for i, t in enumerate(ts)}
        self._coef_table = coef

    @property
    def _clip(self):
        return float(self.config["clip_sample_range"]) if self.config["clip_sample"] else 0.0

    def step(self, model_output, timestep, sample, eta=0.0, **kw):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' first")
        if eta != 0.0:
            raise NotImplementedError("eta != 0")
        _lib.require_gpu(sample, "DDIMScheduler.step(sample)")
        c = self._coef[int(timestep)]
        x = sample.detach().to(torch.float32).contiguous().clone()
        eps = model_output.detach().to(torch.float32).contiguous()
        rc = _lib.load().bndm_ddim_step(_ptr(x), _ptr(eps), float(c[1]), float(c[2]), float(c[3]), float(c[4]),
                                        self._clip, x.numel(), _lib.current_stream_ptr())
        _lib.check(rc, "bndm_ddim_step")
        return SchedulerOutput(prev_sample=x)

    @torch.no_grad()
    def sample(self, model, x):
        """for t in timesteps: x = step(model(x, t).sample, t, x).prev_sample  (ddim_diffusers.py:674-681)."""
        core = unwrap(model)
        if not isinstance(core, UNet2DModel):
            for t in self.timesteps:
                x = self.step(model(x, t).sample, t, x).prev_sample
            return x
        x = x.detach().to(torch.float32).contiguous().clone()
        B = x.shape[0]
        h = core._ensure_engine(B, x.shape[-1], x.device)
        coef = np.ascontiguousarray(self._coef_table, dtype=np.float32)
        rc = _lib.load().bndm_unet_sample_ddim(h, _ptr(x), B, len(coef), coef.ctypes.data_as(C.c_void_p), self._clip,
                                               _lib.current_stream_ptr())
        _lib.check(rc, "bndm_unet_sample_ddim")
        return x


class IADBScheduler(_ConfigIO):
    """latent_iadb_bn_diffusers.py:75-142.  The reference reads noise_type / out_channels from the
    script's global args; here they are constructor arguments."""

    def __init__(self, num_train_timesteps=1000, noise_type="gaussian", out_channels=4):
        self.config = dict(num_train_timesteps=num_train_timesteps, noise_type=noise_type, out_channels=out_channels)
        self.num_train_timesteps = num_train_timesteps
        self.noise_type, self.out_channels = noise_type, out_channels
        self.num_inference_steps = None

    def set_timesteps(self, num_inference_steps):
        self.num_inference_steps = num_inference_steps

    def _coeffs(self, timestep):
        # the reference forms alpha = (t+1)/n and alpha_next = t/n as Python floats (double precision) and multiplies
        # the fp32 tensor by their difference (:95-117): one rounding, of the double difference
        n = self.num_inference_steps
        a, an = (timestep + 1) / n, timestep / n
        return np.float32(a - an), np.float32(a - an)       # gamma == alpha here (:95-99)

    def step(self, model_output, timestep, x_alpha):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after "
                             "creating the scheduler")
        _lib.require_gpu(x_alpha, "IADBScheduler.step(x_alpha)")
        Cc, Cout = x_alpha.shape[1], model_output.shape[1]
        if self.noise_type in ("gaussianBN", "gaussianRN"):
            if self.out_channels not in (Cc, 2 * Cc):
                raise NotImplementedError
        elif self.noise_type != "gaussian":
            raise NotImplementedError
        da, dg = self._coeffs(int(timestep))
        x = x_alpha.detach().to(torch.float32).contiguous().clone()
        d = model_output.detach().to(torch.float32).contiguous()
        rc = _lib.load().bndm_iadb_step(_ptr(x), _ptr(d), float(da), float(dg), x.shape[0], Cc, Cout,
                                        x.shape[2] * x.shape[3], _lib.current_stream_ptr())
        _lib.check(rc, "bndm_iadb_step")
        return x

    def add_noise(self, original_samples, noise, alpha):
        """(1 - alpha) * original + alpha * noise (latent_iadb_bn_diffusers.py:128-139), one HIP kernel."""
        from .training import train_targets
        return train_targets(noise, original_samples, None, None, alpha, None)[0]

    def __len__(self):
        return self.num_train_timesteps

    @torch.no_grad()
    def sample(self, model, x):
        """the loop of latent_iadb_bn_diffusers.py:524-529 inside the engine."""
        from .sampler import _iadb_loop
        n = self.num_inference_steps
        oc = self.out_channels
        nt = self.noise_type
        # step tables in double precision, as step() forms them: bit-identical to calling step() n times
        t_in = np.array([(n - 1 - s + 1) / n for s in range(n)], dtype=np.float32)
        d = np.array([np.float32((n - 1 - s + 1) / n - (n - 1 - s) / n) for s in range(n)], dtype=np.float32)
        out, _, _ = _iadb_loop(model, x, None, n, "linear", "linear", (1.0, 0.0, 3.0), oc, nt, "train", 1,
                               tables=(t_in, d, d.copy()))
        return out
