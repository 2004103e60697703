"""Decoder half of ``AutoencoderKL`` on the HIP engine.

The reference decodes sampled latents with ``vae = AutoencoderKL.from_pretrained("stabilityai/sd-vae-ft-mse")`` and
``vae.decode((x / 0.18215).half()).sample`` (latent_iadb_bn_diffusers.py:70-71,185-191,531-533).  This module keeps
that call surface -- ``AutoencoderKL(...)``, ``.decode(z).sample``, ``load_state_dict`` with diffusers' key names
(``post_quant_conv.*``, ``decoder.*``; encoder keys of a full checkpoint are ignored), ``from_pretrained`` of a local
directory -- and hands the latent to ``bndm_vae_decode`` (csrc/unet_engine.hip: post_quant_conv, conv_in, mid block
with a one-head attention over all latent pixels, four up blocks, GroupNorm + SiLU + conv_out; 16-bit storage with
fp32 accumulation like the reference's ``.half()`` model).  There is no PyTorch compute path behind it and the
checkpoint is not available offline: weights are whatever the caller loads (tests use seeded synthetic ones).
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass

import torch
from torch import nn

from . import _lib
from .unet import DTYPE_BF16, DTYPE_F16, _Attn, _conv, _norm

SCALING_FACTOR = 0.18215


@dataclass
class DecoderOutput:
    sample: torch.Tensor


class _VResnet(nn.Module):
    def __init__(self, ci, co):
        super().__init__()
        self.norm1 = _norm(ci)
        self.conv1 = _conv(ci, co, 3)
        self.norm2 = _norm(co)
        self.conv2 = _conv(co, co, 3)
        if ci != co:
            self.conv_shortcut = _conv(ci, co, 1)


class _Up(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = _conv(c, c, 3)


class _UpBlock(nn.Module):
    def __init__(self, ci, co, layers, upsample):
        super().__init__()
        self.resnets = nn.ModuleList([_VResnet(ci if j == 0 else co, co) for j in range(layers)])
        if upsample:
            self.upsamplers = nn.ModuleList([_Up(co)])


class _Mid(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.resnets = nn.ModuleList([_VResnet(c, c), _VResnet(c, c)])
        self.attentions = nn.ModuleList([_Attn(c)])


class _Decoder(nn.Module):
    def __init__(self, latent_channels, out_channels, boc, layers_per_block):
        super().__init__()
        rev = tuple(reversed(boc))
        self.conv_in = _conv(latent_channels, rev[0], 3)
        self.mid_block = _Mid(rev[0])
        blocks, prev = [], rev[0]
        for i, oc in enumerate(rev):
            blocks.append(_UpBlock(prev, oc, layers_per_block + 1, i != len(rev) - 1))
            prev = oc
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = _norm(rev[-1])
        self.conv_out = _conv(rev[-1], out_channels, 3)


class AutoencoderKL(nn.Module):
    """Decoder-only ``AutoencoderKL``: constructor arguments follow diffusers (defaults = sd-vae-ft-mse)."""

    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 latent_channels=4, norm_num_groups=32, act_fn="silu", scaling_factor=SCALING_FACTOR, sample_size=512,
                 dtype="f16", seed=0, **unused):
        super().__init__()
        if norm_num_groups != 32 or act_fn != "silu":
            raise NotImplementedError("AutoencoderKL: norm_num_groups=32 and act_fn='silu' only")
        self.config = dict(in_channels=in_channels, out_channels=out_channels,
                           block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                           latent_channels=latent_channels, norm_num_groups=32, act_fn="silu",
                           scaling_factor=scaling_factor, sample_size=sample_size)
        self.compute_dtype = dtype
        self.post_quant_conv = _conv(latent_channels, latent_channels, 1)
        self.decoder = _Decoder(latent_channels, out_channels, tuple(block_out_channels), layers_per_block)
        gen = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for m in self.modules():
                if hasattr(m, "reset"):
                    m.reset(gen)
        self._engine = None
        self._engine_key = None

    # checkpoints of the full autoencoder also carry encoder.* / quant_conv.*: not needed for decode
    def load_state_dict(self, state_dict, strict=True):
        from .unet import convert_deprecated_attention_keys
        sd = {k: v for k, v in state_dict.items() if k.startswith(("decoder.", "post_quant_conv."))}
        return super().load_state_dict(convert_deprecated_attention_keys(sd), strict=strict)

    def half(self):                      # vae.half() in user code: storage is 16-bit inside the engine already
        return self

    def release_engine(self):
        if self._engine is not None:
            _lib.load().bndm_unet_destroy(self._engine)
            self._engine = None
            self._engine_key = None

    def __del__(self):
        try:
            self.release_engine()
        except Exception:
            pass

    def _ensure_engine(self, B, res, device):
        ver = tuple(p._version for p in self.parameters()) + tuple(p.data_ptr() for p in self.parameters())
        key = (ver, res, self.compute_dtype, device.index)
        if self._engine is not None and self._engine_key[0] == key and self._engine_key[1] >= B:
            return self._engine
        self.release_engine()
        lib = _lib.load()
        cfg = _lib.VaeConfig()
        c = self.config
        cfg.latent_channels, cfg.out_channels, cfg.latent_resolution = c["latent_channels"], c["out_channels"], res
        cfg.num_levels = len(c["block_out_channels"])
        for i, v in enumerate(c["block_out_channels"]):
            cfg.block_out_channels[i] = v
        cfg.layers_per_block = c["layers_per_block"]
        cfg.dtype = {"f16": DTYPE_F16, "fp16": DTYPE_F16, "bf16": DTYPE_BF16}[self.compute_dtype]
        cfg.max_batch = max(B, 1)
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.bndm_vae_decoder_create(C.byref(h), C.byref(cfg)), "bndm_vae_decoder_create")
            try:
                sd = self.state_dict()
                name = C.create_string_buffer(200)
                numel = C.c_int64()
                for i in range(lib.bndm_unet_num_params(h)):
                    _lib.check(lib.bndm_unet_param_info(h, i, name, 200, C.byref(numel)), "param_info")
                    k = name.value.decode()
                    if k not in sd:
                        raise KeyError(f"state dict lacks '{k}'")
                    t = sd[k].detach().to("cpu", torch.float32).contiguous()
                    _lib.check(lib.bndm_unet_load_param(h, name.value, C.c_void_p(t.data_ptr()), t.numel()),
                               f"load_param({k})")
                _lib.check(lib.bndm_unet_finalize(h), "bndm_unet_finalize")
            except Exception:
                lib.bndm_unet_destroy(h)
                raise
        self._engine, self._engine_key = h, (key, cfg.max_batch)
        return h

    @torch.no_grad()
    def decode(self, z, return_dict=True):
        """``vae.decode(z)``: z [B, latent_channels, r, r] on the GPU (any float dtype), already divided by the
        scaling factor by the caller as in the reference.  Returns fp32 images [B, out_channels, 8r, 8r]."""
        _lib.require_gpu(z, "AutoencoderKL.decode(z)")
        z = z.detach().to(torch.float32).contiguous()
        B, L, H, W = z.shape
        if L != self.config["latent_channels"] or H != W:
            raise ValueError(f"latent shape {tuple(z.shape)} does not fit latent_channels={self.config['latent_channels']}")
        up = 1 << (len(self.config["block_out_channels"]) - 1)
        out = torch.empty((B, self.config["out_channels"], H * up, W * up), dtype=torch.float32, device=z.device)
        lib = _lib.load()
        step = 8                                           # the engine's batch limit (512^2 activations)
        for b0 in range(0, B, step):
            nb = min(step, B - b0)
            h = self._ensure_engine(min(B, step), H, z.device)
            rc = lib.bndm_vae_decode(h, C.c_void_p(z[b0:b0 + nb].data_ptr()), C.c_void_p(out[b0:b0 + nb].data_ptr()),
                                     nb, _lib.current_stream_ptr())
            _lib.check(rc, "bndm_vae_decode")
        return DecoderOutput(sample=out) if return_dict else (out,)

    forward = decode

    def save_pretrained(self, directory, safe_serialization=True):
        os.makedirs(directory, exist_ok=True)
        with open(os.path.join(directory, "config.json"), "w") as f:
            json.dump(dict(self.config, _class_name="AutoencoderKL"), f, indent=2)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(directory, "diffusion_pytorch_model.safetensors"))
        else:
            torch.save(sd, os.path.join(directory, "diffusion_pytorch_model.bin"))

    @classmethod
    def from_pretrained(cls, directory, use_safetensors=True, **kw):
        """Local directory in diffusers layout (the hub id the reference passes cannot be fetched offline)."""
        if not os.path.isdir(directory):
            raise FileNotFoundError(f"AutoencoderKL.from_pretrained: '{directory}' is not a local directory (no network)")
        with open(os.path.join(directory, "config.json")) as f:
            cfg = json.load(f)
        keep = ("in_channels", "out_channels", "block_out_channels", "layers_per_block", "latent_channels",
                "norm_num_groups", "act_fn", "scaling_factor", "sample_size")
        model = cls(**{k: cfg[k] for k in keep if k in cfg}, **kw)
        st = os.path.join(directory, "diffusion_pytorch_model.safetensors")
        if use_safetensors and os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(directory, "diffusion_pytorch_model.bin"), map_location="cpu")
        model.load_state_dict(sd)
        return model


def vae_decode(vae, x):
    """latent_iadb_bn_diffusers.py:185-191: ``latents = 1 / 0.18215 * latents; vae.decode(latents.half()).sample``."""
    return vae.decode((1 / vae.config["scaling_factor"] * x).half()).sample
