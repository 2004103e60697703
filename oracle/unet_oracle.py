"""TEST INFRASTRUCTURE ONLY -- fp32 torch-CPU restatement of diffusers ``UNet2DModel``.

The reference builds the network with ``diffusers.UNet2DModel`` (iadb_bn.py:205-282,
utils.py:7-84, ddim_diffusers.py:377-453, latent_iadb_bn_diffusers.py:337-372) and calls it as
``model(x, t, return_dict=False)[0]`` (iadb_bn.py:319).  ``diffusers`` (PyPI, version unpinned in
the reference: README.md:47; hint ``check_min_version("0.26.0.dev0")`` at ddim_diffusers.py:38)
is NOT vendored under /root/reference and is NOT installed in this image, so this file restates
its published architecture from the call sites' constructor arguments:

PARITY UNPINNED at this boundary: there is no reference output to compare with.  What *is*
pinned: the exact parameter count of the res64 3->3 network (113,673,219 -- the size of the
identical google/ddpm-celebahq-256-style config) and the state-dict key list
(tests/test_oracle_unet.py).

Only tests/, smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

GROUPS = 32
EPS = 1e-5
HEAD_DIM = 8
TEMB_IN = None  # = block_out_channels[0]


def make_config(res: int = 64, in_channels: int = 3, out_channels: int = 3, latent: bool = False):
    """Constructor arguments used by the reference for each resolution.

    iadb_bn.py:209-276 / utils.py:11-79 (pixel space), latent_iadb_bn_diffusers.py:340-361."""
    if latent:
        if res in (64, 512):
            boc, attn_down, attn_up = (128, 128, 256, 256, 512, 512), 4, 1
        elif res == 128:
            boc, attn_down, attn_up = (128, 128, 128, 256, 256, 512, 512), 5, 1
        elif res == 256:
            boc, attn_down, attn_up = (128, 256, 256), 2, 0
        else:
            raise ValueError(f"Unsupported resolution: {res}")
    else:
        if res == 64:
            boc, attn_down, attn_up = (128, 128, 256, 256, 512, 512), 4, 1
        elif res == 128:
            boc, attn_down, attn_up = (128, 128, 128, 256, 256, 512, 512), 5, 1
        elif res == 256:
            boc, attn_down, attn_up = (128, 128, 128, 128, 256, 256, 512, 512), 6, 1
        else:
            raise NotImplementedError(res)
    n = len(boc)
    return dict(
        in_channels=in_channels, out_channels=out_channels, block_out_channels=tuple(boc),
        down_attn=tuple(i == attn_down for i in range(n)),
        up_attn=tuple(i == attn_up for i in range(n)),
        layers_per_block=2,
    )


# ----------------------------------------------------------------------------- parameter spec
def param_shapes(cfg) -> "OrderedDict[str, tuple]":
    """State-dict key -> shape, in diffusers' naming."""
    boc = cfg["block_out_channels"]
    n = len(boc)
    temb = boc[0] * 4
    P = OrderedDict()

    def conv(name, cin, cout, k):
        P[name + ".weight"] = (cout, cin, k, k)
        P[name + ".bias"] = (cout,)

    def lin(name, cin, cout):
        P[name + ".weight"] = (cout, cin)
        P[name + ".bias"] = (cout,)

    def norm(name, c):
        P[name + ".weight"] = (c,)
        P[name + ".bias"] = (c,)

    def resnet(name, cin, cout):
        norm(name + ".norm1", cin)
        conv(name + ".conv1", cin, cout, 3)
        lin(name + ".time_emb_proj", temb, cout)
        norm(name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".conv_shortcut", cin, cout, 1)

    def attn(name, c):
        norm(name + ".group_norm", c)
        lin(name + ".to_q", c, c)
        lin(name + ".to_k", c, c)
        lin(name + ".to_v", c, c)
        lin(name + ".to_out.0", c, c)

    conv("conv_in", cfg["in_channels"], boc[0], 3)
    lin("time_embedding.linear_1", boc[0], temb)
    lin("time_embedding.linear_2", temb, temb)
    out_c = boc[0]
    for i in range(n):
        in_c, out_c = out_c, boc[i]
        for j in range(cfg["layers_per_block"]):
            resnet(f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c)
            if cfg["down_attn"][i]:
                attn(f"down_blocks.{i}.attentions.{j}", out_c)
        if i != n - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", out_c, out_c, 3)
    mid = boc[-1]
    resnet("mid_block.resnets.0", mid, mid)
    attn("mid_block.attentions.0", mid)
    resnet("mid_block.resnets.1", mid, mid)
    rev = list(reversed(boc))
    out_c = rev[0]
    for i in range(n):
        prev, out_c = out_c, rev[i]
        in_c = rev[min(i + 1, n - 1)]
        nl = cfg["layers_per_block"] + 1
        for j in range(nl):
            skip = in_c if j == nl - 1 else out_c
            rin = prev if j == 0 else out_c
            resnet(f"up_blocks.{i}.resnets.{j}", rin + skip, out_c)
            if cfg["up_attn"][i]:
                attn(f"up_blocks.{i}.attentions.{j}", out_c)
        if i != n - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", out_c, out_c, 3)
    norm("conv_norm_out", boc[0])
    conv("conv_out", boc[0], cfg["out_channels"], 3)
    return P


def init_params(cfg, seed: int = 0, perturb_norm: float = 0.0):
    """torch-default init (what diffusers leaves in place): U(+-1/sqrt(fan_in)) for conv/linear
    weights and biases, GroupNorm weight 1 / bias 0.  ``perturb_norm`` > 0 adds N(0, p) to the
    norm affine parameters so that tests exercise them."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    shapes = param_shapes(cfg)
    for name, shp in shapes.items():
        base = name.rsplit(".", 1)[0]
        is_norm = base.endswith(("norm1", "norm2", "group_norm", "conv_norm_out"))
        if is_norm:
            v = torch.ones(shp) if name.endswith("weight") else torch.zeros(shp)
            if perturb_norm:
                v = v + perturb_norm * torch.randn(shp, generator=g)
        else:
            wshape = shapes[base + ".weight"]
            fan_in = 1
            for d in wshape[1:]:
                fan_in *= d
            bound = 1.0 / math.sqrt(fan_in)
            v = (torch.rand(shp, generator=g) * 2 - 1) * bound
        sd[name] = v.float()
    return sd


# ----------------------------------------------------------------------------- forward
def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = t.float()[:, None] * freqs[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def _gn(x, sd, name):
    return F.group_norm(x, GROUPS, sd[name + ".weight"], sd[name + ".bias"], EPS)


def _conv(x, sd, name, stride=1, pad=1):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=pad)


def _resnet(x, emb, sd, name):
    h = _conv(F.silu(_gn(x, sd, name + ".norm1")), sd, name + ".conv1")
    tp = F.linear(F.silu(emb), sd[name + ".time_emb_proj.weight"], sd[name + ".time_emb_proj.bias"])
    h = h + tp[:, :, None, None]
    h = _conv(F.silu(_gn(h, sd, name + ".norm2")), sd, name + ".conv2")
    if (name + ".conv_shortcut.weight") in sd:
        x = _conv(x, sd, name + ".conv_shortcut", pad=0)
    return x + h


def _attn(x, sd, name):
    B, C, H, W = x.shape
    heads = C // HEAD_DIM
    h = _gn(x, sd, name + ".group_norm").view(B, C, H * W).transpose(1, 2)      # [B, T, C]
    q = F.linear(h, sd[name + ".to_q.weight"], sd[name + ".to_q.bias"])
    k = F.linear(h, sd[name + ".to_k.weight"], sd[name + ".to_k.bias"])
    v = F.linear(h, sd[name + ".to_v.weight"], sd[name + ".to_v.bias"])
    sp = lambda z: z.view(B, -1, heads, HEAD_DIM).transpose(1, 2)              # [B, h, T, d]
    q, k, v = sp(q), sp(k), sp(v)
    w = torch.softmax(q @ k.transpose(-1, -2) * (HEAD_DIM ** -0.5), dim=-1)
    o = (w @ v).transpose(1, 2).reshape(B, -1, C)
    o = F.linear(o, sd[name + ".to_out.0.weight"], sd[name + ".to_out.0.bias"])
    return o.transpose(1, 2).reshape(B, C, H, W) + x


@torch.no_grad()
def forward(sd, cfg, sample: torch.Tensor, timestep) -> torch.Tensor:
    """UNet2DModel.forward: sample [B,Cin,H,W] f32, timestep scalar / 0-d / [B] -> [B,Cout,H,W]."""
    boc = cfg["block_out_channels"]
    n = len(boc)
    B = sample.shape[0]
    t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1)
    t = t * torch.ones(B, dtype=torch.float32)
    emb = timestep_embedding(t, boc[0])
    emb = F.linear(emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    emb = F.linear(F.silu(emb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])

    h = _conv(sample, sd, "conv_in")
    skips = [h]
    for i in range(n):
        for j in range(cfg["layers_per_block"]):
            h = _resnet(h, emb, sd, f"down_blocks.{i}.resnets.{j}")
            if cfg["down_attn"][i]:
                h = _attn(h, sd, f"down_blocks.{i}.attentions.{j}")
            skips.append(h)
        if i != n - 1:
            h = _conv(h, sd, f"down_blocks.{i}.downsamplers.0.conv", stride=2)
            skips.append(h)
    h = _resnet(h, emb, sd, "mid_block.resnets.0")
    h = _attn(h, sd, "mid_block.attentions.0")
    h = _resnet(h, emb, sd, "mid_block.resnets.1")
    for i in range(n):
        for j in range(cfg["layers_per_block"] + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = _resnet(h, emb, sd, f"up_blocks.{i}.resnets.{j}")
            if cfg["up_attn"][i]:
                h = _attn(h, sd, f"up_blocks.{i}.attentions.{j}")
        if i != n - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(h, sd, f"up_blocks.{i}.upsamplers.0.conv")
    h = F.silu(_gn(h, sd, "conv_norm_out"))
    return _conv(h, sd, "conv_out")


class OracleUNet:
    """Callable with the diffusers call surface used by the reference loops."""

    def __init__(self, cfg, sd):
        self.cfg, self.sd = cfg, sd

    def __call__(self, sample, timestep, return_dict=False):
        out = forward(self.sd, self.cfg, sample, timestep)
        return (out,)


def flops_per_image(cfg, res: int) -> float:
    """2*MAC over convs, linears on the spatial path and attention."""
    shapes = param_shapes(cfg)
    boc = cfg["block_out_channels"]
    n = len(boc)
    total = 0.0
    # walk the same structure as forward(), tracking resolution
    def conv_f(name, hw):
        co, ci, k, _ = shapes[name + ".weight"]
        return 2.0 * co * ci * k * k * hw

    def res_f(name, hw):
        f = conv_f(name + ".conv1", hw) + conv_f(name + ".conv2", hw)
        if name + ".conv_shortcut.weight" in shapes:
            f += conv_f(name + ".conv_shortcut", hw)
        co, ci = shapes[name + ".time_emb_proj.weight"]
        return f + 2.0 * co * ci

    def attn_f(name, hw):
        c = shapes[name + ".to_q.weight"][0]
        return 4 * 2.0 * c * c * hw + 2 * 2.0 * hw * hw * c

    r = res
    total += conv_f("conv_in", r * r)
    total += 2.0 * (boc[0] * boc[0] * 4 + (boc[0] * 4) ** 2)
    for i in range(n):
        for j in range(cfg["layers_per_block"]):
            total += res_f(f"down_blocks.{i}.resnets.{j}", r * r)
            if cfg["down_attn"][i]:
                total += attn_f(f"down_blocks.{i}.attentions.{j}", r * r)
        if i != n - 1:
            r //= 2
            total += conv_f(f"down_blocks.{i}.downsamplers.0.conv", r * r)
    total += res_f("mid_block.resnets.0", r * r) + res_f("mid_block.resnets.1", r * r)
    total += attn_f("mid_block.attentions.0", r * r)
    for i in range(n):
        for j in range(cfg["layers_per_block"] + 1):
            total += res_f(f"up_blocks.{i}.resnets.{j}", r * r)
            if cfg["up_attn"][i]:
                total += attn_f(f"up_blocks.{i}.attentions.{j}", r * r)
        if i != n - 1:
            r *= 2
            total += conv_f(f"up_blocks.{i}.upsamplers.0.conv", r * r)
    total += conv_f("conv_out", r * r)
    return total
