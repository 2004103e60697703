"""``accelerate launch latent_iadb_bn_diffusers.py --train_or_test=test ...`` on MI355X: the test branch
of the reference's latent IADB script (latent_iadb_bn_diffusers.py:472-574; BNDM flags of input_args.py
:217-229, the HF training flags are accepted and ignored).

The latent loop (4-channel 64x64 latents, UNet 4 -> 4 or 4 -> 8, gamma == alpha linear,
:84-122,:524-529) runs inside the HIP engine.  The reference then decodes with the
``stabilityai/sd-vae-ft-mse`` VAE fetched from the HF hub (:70,:185-191,:531-540).  The hub is unreachable here:
``--vae_dir`` takes a local directory in diffusers layout (config.json + diffusion_pytorch_model.safetensors);
without it the decoder (bndm_amd/vae.py, HIP) runs with seeded random-init weights, which exercises the whole
path but does not produce pictures.  Outputs: ``<output_dir>/images/<name>_<idx>.png`` as the reference writes them
(:540,:566), plus the final latents ``<output_dir>/latents/<name>_<idx>.npy``.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

_LEVELS = {   # latent_iadb_bn_diffusers.py:340-361, keyed by image resolution
    64: ((128, 128, 256, 256, 512, 512), 4, 1), 512: ((128, 128, 256, 256, 512, 512), 4, 1),
    128: ((128, 128, 128, 256, 256, 512, 512), 5, 1), 256: ((128, 256, 256), 2, 0),
}


def build_parser():
    p = argparse.ArgumentParser(description="bndm latent IADB sampling on MI355X")
    a = p.add_argument
    a("--dataset_name", type=str, default=None)
    a("--resolution", type=int, default=64)
    a("--train_or_test", type=str, default="train")
    a("--test_samples", type=int, default=10)
    a("--eval_batch_size", type=int, default=16)
    a("--output_dir", type=str, default="ddpm-model-64")
    a("--ddpm_num_inference_steps", type=int, default=250)
    a("--out_channels", type=int, default=4)
    a("--noise_type", type=str, default="gaussian")
    a("--seed", type=int, default=0)
    a("--use_ema", action="store_true")
    for flag, typ in (("train_batch_size", int), ("num_epochs", int), ("gradient_accumulation_steps", int),
                      ("learning_rate", float), ("lr_warmup_steps", int), ("lr_scheduler", str),
                      ("mixed_precision", str), ("dataloader_num_workers", int), ("save_images_epochs", int),
                      ("save_model_epochs", int), ("checkpointing_steps", int), ("logger", str),
                      ("logging_dir", str), ("train_data_dir", str), ("ddpm_num_steps", int),
                      ("ddpm_beta_schedule", str), ("prediction_type", str), ("model_config_name_or_path", str),
                      ("cache_dir", str), ("resume_from_checkpoint", str), ("local_rank", int)):
        a(f"--{flag}", type=typ, default=None, help=argparse.SUPPRESS)
    for flag in ("random_flip", "center_crop", "overwrite_output_dir", "push_to_hub"):
        a(f"--{flag}", action="store_true", help=argparse.SUPPRESS)
    g = p.add_argument_group("MI355X build additions")
    g.add_argument("--full_batches", action="store_true", help="sample every latent of every batch")
    g.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    g.add_argument("--root", default=".")
    g.add_argument("--vae_dir", default=None, help="local AutoencoderKL directory (diffusers layout)")
    g.add_argument("--no_decode", action="store_true", help="write latents only")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    from . import _lib
    from .parallel import init_from_env, shard_range
    from .schedulers import IADBScheduler
    from .unet import UNet2DModel

    rank, world, local = init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("latent_iadb_bn_diffusers.py (MI355X build): no GPU visible; no CPU fallback")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    _lib.load()
    say = print if rank == 0 else (lambda *a, **k: None)
    os.chdir(args.root)
    if args.train_or_test != "test":
        raise SystemExit("training is outside the scope of the MI355X sampling build; use --train_or_test=test")
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)

    out_dir = os.path.join("results_gaussianBN", f"{args.output_dir}_{args.noise_type}" + ("_ema" if args.use_ema else ""))
    out_channels = args.out_channels * 2 if args.noise_type in ("gaussianBN", "gaussianRN") else args.out_channels  # :282-283
    if rank == 0:
        os.makedirs(os.path.join(out_dir, "latents"), exist_ok=True)
        os.makedirs(os.path.join(out_dir, "images"), exist_ok=True)
    say("===> Start testing!")
    scheduler = IADBScheduler(noise_type=args.noise_type, out_channels=out_channels)
    scheduler.set_timesteps(args.ddpm_num_inference_steps)
    udir = os.path.join(out_dir, "unet")
    if os.path.exists(os.path.join(udir, "config.json")):
        model = UNet2DModel.from_pretrained(udir, use_safetensors=True, dtype=args.dtype)
    else:
        if args.resolution not in _LEVELS:
            raise ValueError(f"Unsupported resolution: {args.resolution}")
        boc, ad, au = _LEVELS[args.resolution]
        n = len(boc)
        say(f"[bndm] {udir} not found: sampling from seeded random-init weights")
        model = UNet2DModel(sample_size=args.resolution, in_channels=4, out_channels=out_channels, layers_per_block=2,
                            block_out_channels=boc,
                            down_block_types=tuple("AttnDownBlock2D" if i == ad else "DownBlock2D" for i in range(n)),
                            up_block_types=tuple("AttnUpBlock2D" if i == au else "UpBlock2D" for i in range(n)),
                            dtype=args.dtype, seed=args.seed)
    model = model.to(device).eval()
    vae = None
    lat = args.resolution // 8
    if not args.no_decode and lat in (16, 32, 64):
        from .vae import AutoencoderKL, vae_decode
        if args.vae_dir:
            vae = AutoencoderKL.from_pretrained(args.vae_dir, dtype=args.dtype)
        else:
            say("[bndm] no --vae_dir (stabilityai/sd-vae-ft-mse lives on the HF hub): decoding with seeded random-init "
                "VAE weights")
            vae = AutoencoderKL(dtype=args.dtype, seed=args.seed)
        vae = vae.to(device).eval()
    else:
        say("[bndm] VAE decode skipped: writing latents + previews")
    name = {"gaussian": "iadb_gwn", "gaussianBN": "iadb_gwn2gbn"}.get(args.noise_type)
    if name is None:
        raise ValueError(f"Unsupported noise type: {args.noise_type}")
    cnt = 0
    for i in range(args.test_samples // args.eval_batch_size):
        noise = np.random.randn(args.eval_batch_size, 4, lat, lat).astype(np.float32)       # white x0 (:502)
        if not args.full_batches:                                                          # figure-9 picks (:505-513)
            if i == 0:
                noise = noise[[2, 7, 31, 48]] if args.eval_batch_size > 48 else noise[:1]
            elif i == 1:
                noise = noise[[6]] if args.eval_batch_size > 6 else noise[:1]
            else:
                continue
        B = noise.shape[0]
        b0, bc = shard_range(B, rank, world)
        if bc == 0:
            continue
        x = scheduler.sample(model, torch.from_numpy(noise[b0:b0 + bc]).to(device))        # loop of :524-529
        z = (x / 0.18215).cpu().numpy()                                                    # what vae.decode gets (:186)
        images = None
        if vae is not None:
            from .sampler import export_u8
            images = export_u8(vae_decode(vae, x), "round").cpu().numpy()                  # :539-540
        from PIL import Image
        for j in range(bc):
            idx = cnt + b0 + j + 1
            np.save(os.path.join(out_dir, "latents", f"{name}_{idx:05d}.npy"), z[j])
            if images is not None:
                Image.fromarray(images[j]).save(os.path.join(out_dir, "images", f"{name}_{idx:05d}.png"))   # :566
                continue
            pv = z[j, :3]
            pv = (pv - pv.min()) / max(pv.max() - pv.min(), 1e-8)
            Image.fromarray((pv.transpose(1, 2, 0) * 255).astype(np.uint8)).save(
                os.path.join(out_dir, "images", f"{name}_{idx:05d}_latent_preview.png"))
        cnt += B
    say("Done.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
