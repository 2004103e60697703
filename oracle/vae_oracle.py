"""TEST INFRASTRUCTURE ONLY -- fp32 torch-CPU restatement of the decoder half of diffusers ``AutoencoderKL``.

The reference decodes latents with ``vae = AutoencoderKL.from_pretrained("stabilityai/sd-vae-ft-mse")`` and
``vae.decode((x / 0.18215).half()).sample`` (latent_iadb_bn_diffusers.py:70-71,185-191,531-533).  Neither
``diffusers`` (un-vendored PyPI dependency, version unpinned: README.md:47) nor the checkpoint (HF hub) is available
here, so this file restates the published architecture of that model's decoder from its public config
(block_out_channels (128, 256, 512, 512), layers_per_block 2, latent_channels 4, norm_num_groups 32, act silu,
scaling_factor 0.18215):

    z -> post_quant_conv 1x1 (4->4) -> conv_in 3x3 (4->512)
      -> mid: Resnet(512), Attention(1 head of 512, GN eps 1e-6, residual), Resnet(512)
      -> up 0: 3 x Resnet(512->512), nearest-2x + conv3x3      up 1: the same
      -> up 2: Resnet(512->256 with 1x1 shortcut), 2 x Resnet(256), nearest-2x + conv3x3
      -> up 3: Resnet(256->128 with 1x1 shortcut), 2 x Resnet(128)
      -> GN(32, eps 1e-6) -> SiLU -> conv_out 3x3 (128->3)
    Resnet: GN(32, 1e-6) -> SiLU -> conv3x3 -> GN -> SiLU -> conv3x3, + (shortcut(x) | x); no time embedding.

PARITY UNPINNED: there is no reference output or checkpoint to compare with.  Pinned structurally only: the
state-dict key list and the parameter count of this description (tests/test_oracle_vae.py).

Only tests/ may import this file.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

GROUPS = 32
EPS = 1e-6
SCALING = 0.18215


def make_config(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, out_channels=3):
    return dict(block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                latent_channels=latent_channels, out_channels=out_channels)


def _plan(cfg):
    """[(block index, [(resnet in, out)], upsample?)] of the decoder's up path."""
    rev = tuple(reversed(cfg["block_out_channels"]))
    plan, prev = [], rev[0]
    for i, oc in enumerate(rev):
        res = []
        for j in range(cfg["layers_per_block"] + 1):
            res.append((prev if j == 0 else oc, oc))
        plan.append((i, res, i != len(rev) - 1))
        prev = oc
    return rev, plan


def param_shapes(cfg) -> "OrderedDict[str, tuple]":
    s = OrderedDict()

    def conv(n, ci, co, k):
        s[n + ".weight"] = (co, ci, k, k)
        s[n + ".bias"] = (co,)

    def lin(n, ci, co):
        s[n + ".weight"] = (co, ci)
        s[n + ".bias"] = (co,)

    def norm(n, c):
        s[n + ".weight"] = (c,)
        s[n + ".bias"] = (c,)

    def resnet(n, ci, co):
        norm(n + ".norm1", ci)
        conv(n + ".conv1", ci, co, 3)
        norm(n + ".norm2", co)
        conv(n + ".conv2", co, co, 3)
        if ci != co:
            conv(n + ".conv_shortcut", ci, co, 1)

    L = cfg["latent_channels"]
    rev, plan = _plan(cfg)
    conv("post_quant_conv", L, L, 1)
    conv("decoder.conv_in", L, rev[0], 3)
    resnet("decoder.mid_block.resnets.0", rev[0], rev[0])
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", rev[0])
    for p in ("to_q", "to_k", "to_v", "to_out.0"):
        lin(a + "." + p, rev[0], rev[0])
    resnet("decoder.mid_block.resnets.1", rev[0], rev[0])
    for i, res, up in plan:
        for j, (ci, co) in enumerate(res):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", ci, co)
        if up:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", res[-1][1], res[-1][1], 3)
    norm("decoder.conv_norm_out", rev[-1])
    conv("decoder.conv_out", rev[-1], cfg["out_channels"], 3)
    return s


def num_params(cfg) -> int:
    n = 0
    for shp in param_shapes(cfg).values():
        k = 1
        for d in shp:
            k *= d
        n += k
    return n


def init_params(cfg, seed: int = 0, perturb_norm: float = 0.0):
    """torch-default init: U(+-1/sqrt(fan_in)) for conv/linear weights and biases, GroupNorm 1 / 0 (+ optional noise)."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    shapes = param_shapes(cfg)
    for name, shp in shapes.items():
        base = name.rsplit(".", 1)[0]
        if base.endswith(("norm1", "norm2", "group_norm", "conv_norm_out")):
            v = torch.ones(shp) if name.endswith("weight") else torch.zeros(shp)
            if perturb_norm:
                v = v + perturb_norm * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in shapes[base + ".weight"][1:]:
                fan_in *= d
            v = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(fan_in)
        sd[name] = v.float()
    return sd


def _gn(x, sd, n):
    return F.group_norm(x, GROUPS, sd[n + ".weight"], sd[n + ".bias"], EPS)


def _conv(x, sd, n, pad=1):
    return F.conv2d(x, sd[n + ".weight"], sd[n + ".bias"], padding=pad)


def _resnet(x, sd, n):
    h = _conv(F.silu(_gn(x, sd, n + ".norm1")), sd, n + ".conv1")
    h = _conv(F.silu(_gn(h, sd, n + ".norm2")), sd, n + ".conv2")
    if (n + ".conv_shortcut.weight") in sd:
        x = _conv(x, sd, n + ".conv_shortcut", pad=0)
    return x + h


def _attn(x, sd, n):
    B, C, H, W = x.shape
    h = _gn(x, sd, n + ".group_norm").view(B, C, H * W).transpose(1, 2)          # [B, T, C], one head of dim C
    q = F.linear(h, sd[n + ".to_q.weight"], sd[n + ".to_q.bias"])
    k = F.linear(h, sd[n + ".to_k.weight"], sd[n + ".to_k.bias"])
    v = F.linear(h, sd[n + ".to_v.weight"], sd[n + ".to_v.bias"])
    w = torch.softmax(q @ k.transpose(-1, -2) * (C ** -0.5), dim=-1)
    o = F.linear(w @ v, sd[n + ".to_out.0.weight"], sd[n + ".to_out.0.bias"])
    return o.transpose(1, 2).reshape(B, C, H, W) + x


@torch.no_grad()
def decode(sd, cfg, z: torch.Tensor) -> torch.Tensor:
    """AutoencoderKL.decode(z).sample for a latent that has already been divided by the scaling factor."""
    _, plan = _plan(cfg)
    x = _conv(z.float(), sd, "post_quant_conv", pad=0)
    x = _conv(x, sd, "decoder.conv_in")
    x = _resnet(x, sd, "decoder.mid_block.resnets.0")
    x = _attn(x, sd, "decoder.mid_block.attentions.0")
    x = _resnet(x, sd, "decoder.mid_block.resnets.1")
    for i, res, up in plan:
        for j in range(len(res)):
            x = _resnet(x, sd, f"decoder.up_blocks.{i}.resnets.{j}")
        if up:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(x, sd, f"decoder.up_blocks.{i}.upsamplers.0.conv")
    x = F.silu(_gn(x, sd, "decoder.conv_norm_out"))
    return _conv(x, sd, "decoder.conv_out")


def vae_decode(sd, cfg, x: torch.Tensor) -> torch.Tensor:
    """latent_iadb_bn_diffusers.py:185-191: decode(1 / 0.18215 * x) (the reference's .half() cast is the product
    path's 16-bit storage; the oracle stays in fp32)."""
    return decode(sd, cfg, 1 / SCALING * x)


def flops_per_image(cfg, latent_res: int) -> float:
    """2*MAC of convs, linears and the attention matmuls for one latent of latent_res x latent_res."""
    rev, plan = _plan(cfg)
    L = cfg["latent_channels"]
    r = latent_res
    f = 2.0 * r * r * (L * L + 9 * L * rev[0])

    def resnet(ci, co, rr):
        return 2.0 * rr * rr * (9 * ci * co + 9 * co * co + (ci * co if ci != co else 0))

    f += 2 * resnet(rev[0], rev[0], r)
    T = r * r
    f += 2.0 * T * 4 * rev[0] * rev[0] + 2.0 * 2 * T * T * rev[0]
    for i, res, up in plan:
        for ci, co in res:
            f += resnet(ci, co, r)
        if up:
            r *= 2
            f += 2.0 * r * r * 9 * res[-1][1] * res[-1][1]
    f += 2.0 * r * r * 9 * rev[-1] * cfg["out_channels"]
    return f
