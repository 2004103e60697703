#!/usr/bin/env python
"""Pin the network / DDIM / VAE oracles against the REAL third-party dependency, the day it is importable.

The reference takes its network, its DDIM step and its VAE decoder from ``diffusers`` (un-vendored PyPI dependency,
version unpinned -- README.md:47; call sites iadb_bn.py:282,319, ddim_diffusers.py:499-503,680,
latent_iadb_bn_diffusers.py:70,188).  ``diffusers`` is not installed in the build image and cannot be fetched
(no network), so oracle/unet_oracle.py, oracle/vae_oracle.py and oracle/sampler_oracle.py::ddim_* are restatements
whose parity is UNPINNED.  This script closes that gap wherever a diffusers wheel exists:

    python tests/golden/make_diffusers_golden.py          # writes tests/golden/diffusers_cases.npz

It builds the real ``UNet2DModel`` / ``DDIMScheduler`` / ``AutoencoderKL`` with the reference's constructor arguments,
loads the SAME seeded synthetic weights the oracle uses (oracle.*.init_params -- diffusers' own state-dict key names,
``strict=True``: a key or shape mismatch fails here, which is itself a pin), runs them in fp32 on CPU on seeded inputs
and stores strided subsamples + float64 sums of the outputs.  Inputs and weights are regenerated from seeds on the
consumer side (tests/test_oracle_diffusers.py), so the fixture is data only -- a few hundred KB.  Without diffusers
the script exits with a message and the consumers skip ("parity unpinned" stays in DESIGN.md section 2).

CASES below is shared with the consumers (imported from this file): keep both sides in step.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "diffusers_cases.npz")
STRIDE = 7

# (name, res the layout is named after, in, out, latent layout?, input H=W, batch, timestep values)
UNET_CASES = [
    ("res64_3to6", 64, 3, 6, False, 64, 2, (0.25, 0.9)),             # iadb_bn.py:209-228, gaussianBN (out 6)
    ("res64_3to3_ddim", 64, 3, 3, False, 64, 1, (990,)),             # ddim_diffusers.py:679: integer timestep
    ("res128_3to6", 128, 3, 6, False, 128, 1, (0.5,)),               # iadb_bn.py:230-251
    ("latent64_4to8", 64, 4, 8, True, 64, 1, (0.004,)),              # latent_iadb_bn_diffusers.py:337-341
    ("latent_celeba256_4to8", 256, 4, 8, True, 32, 2, (0.3, 1.0)),   # latent_iadb_bn_diffusers.py:352-357
]
DDIM_STEPS = (100, 250)
VAE_CASE = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent=16, batch=1)


def unet_inputs(name, cin, hw, batch, tvals, seed=11):
    g = torch.Generator().manual_seed(seed + sum(map(ord, name)))
    x = torch.randn(batch, cin, hw, hw, generator=g)
    integer = all(float(v).is_integer() and v > 1 for v in tvals)
    t = torch.tensor(tvals, dtype=torch.int64 if integer else torch.float32)
    return x, t


def ddim_inputs(seed=5):
    g = torch.Generator().manual_seed(seed)
    return 1.5 * torch.randn(2, 3, 16, 16, generator=g), torch.randn(2, 3, 16, 16, generator=g)


def vae_input(seed=9):
    g = torch.Generator().manual_seed(seed)
    return 0.18215 * torch.randn(VAE_CASE["batch"], 4, VAE_CASE["latent"], VAE_CASE["latent"], generator=g)


def summarize(out, key, y):
    y = y.detach().to(torch.float64).contiguous().view(-1)
    out[key + "/sub"] = y[::STRIDE].to(torch.float32).numpy()
    out[key + "/sum"] = np.array([float(y.sum()), float((y * y).sum())])
    out[key + "/shape"] = np.array(y.numel())


def main():
    try:
        import diffusers
        from diffusers import AutoencoderKL, DDIMScheduler, UNet2DModel
    except Exception as e:                                   # noqa: BLE001 -- any import failure means "not here"
        print(f"diffusers is not importable here ({e!r}): nothing written; the oracles stay 'parity unpinned'.")
        return 2
    sys.path.insert(0, ROOT)
    from oracle import unet_oracle as U
    from oracle import vae_oracle as V
    out = {"diffusers_version": np.array(diffusers.__version__)}
    torch.set_grad_enabled(False)

    for name, res, cin, cout, latent, hw, batch, tvals in UNET_CASES:
        cfg = U.make_config(res, cin, cout, latent=latent)
        sd = U.init_params(cfg, seed=21, perturb_norm=0.1)
        kw = dict(in_channels=cin, out_channels=cout, block_out_channels=cfg["block_out_channels"],
                  down_block_types=tuple("AttnDownBlock2D" if a else "DownBlock2D" for a in cfg["down_attn"]),
                  up_block_types=tuple("AttnUpBlock2D" if a else "UpBlock2D" for a in cfg["up_attn"]))
        if latent:
            kw.update(sample_size=res, layers_per_block=2)               # latent_iadb_bn_diffusers.py:364-372
        else:
            kw.update(act_fn="silu", add_attention=True)                 # iadb_bn.py:282
        m = UNet2DModel(**kw).eval()
        m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
        x, t = unet_inputs(name, cin, hw, batch, tvals)
        # the reference passes a [B] float tensor (iadb_bn.py:319) or a 0-d / python int (ddim_diffusers.py:679)
        y = m(x, t if t.numel() > 1 else t[0], return_dict=False)[0]
        summarize(out, "unet/" + name, y)
        print(f"unet {name}: out {tuple(y.shape)}  |y| {float(y.norm()):.6g}")

    sch = DDIMScheduler(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, beta_schedule="linear")  # ddim_diffusers.py:499-503
    x, eps = ddim_inputs()
    for n in DDIM_STEPS:
        sch.set_timesteps(n)
        ts = [int(v) for v in sch.timesteps]
        out[f"ddim/{n}/timesteps"] = np.array(ts, dtype=np.int64)
        for t in (ts[0], ts[len(ts) // 2], ts[-1]):
            summarize(out, f"ddim/{n}/step{t}", sch.step(eps, t, x).prev_sample)

    vae = AutoencoderKL(in_channels=3, out_channels=3, latent_channels=4,
                        down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4,
                        block_out_channels=VAE_CASE["block_out_channels"], layers_per_block=VAE_CASE["layers_per_block"],
                        norm_num_groups=32, act_fn="silu", sample_size=256).eval()    # sd-vae-ft-mse's public config
    vcfg = V.make_config(VAE_CASE["block_out_channels"], VAE_CASE["layers_per_block"])
    vsd = V.init_params(vcfg, seed=22, perturb_norm=0.1)
    missing, unexpected = vae.load_state_dict({k: v.clone() for k, v in vsd.items()}, strict=False)
    assert not unexpected, unexpected                        # every decoder key of the oracle exists in diffusers
    assert all(k.startswith(("encoder.", "quant_conv.")) for k in missing), missing   # only the unused encoder half
    lat = vae_input()
    summarize(out, "vae/decode", vae.decode(lat / 0.18215).sample)     # latent_iadb_bn_diffusers.py:188
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({os.path.getsize(OUT) / 1024:.0f} KiB) from diffusers {diffusers.__version__}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
