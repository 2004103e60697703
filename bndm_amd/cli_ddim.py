"""``accelerate launch ddim_diffusers.py --train_or_test=test ...`` on MI355X: the test branch of the
reference's DDIM baseline (ddim_diffusers.py:624-712; flags :61-282, the ones the sampling scripts pass
are honoured, HF training flags are accepted and ignored).

As in the reference this is plain white-noise DDIM (no blue noise, no gamma schedule; SURVEY.md
section 0 item 5).  Additions: ``--full_batches`` (no replicability clamps), ``--dtype``,
synthetic weights when ``<output_dir>/unet`` is absent, batch sharding under torch.distributed.run.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

REPLICABILITY_BATCHES = {            # ddim_diffusers.py:655-664
    "cat_res64": [4], "cat_res128": [0, 52], "celeba_res64": [37], "celeba_res128": [10, 26],
    "church_res64": [4, 23, 32, 36],
}

_LEVELS = {64: ((128, 128, 256, 256, 512, 512), 4), 128: ((128, 128, 128, 256, 256, 512, 512), 5),
           256: ((128, 128, 128, 256, 256, 512, 512), 5)}    # ddim_diffusers.py:383-449


def build_parser():
    p = argparse.ArgumentParser(description="bndm DDIM baseline sampling on MI355X (drop-in for ddim_diffusers.py)")
    a = p.add_argument
    a("--dataset_name", type=str, default=None)
    a("--resolution", type=int, default=64)
    a("--train_or_test", type=str, default="train")
    a("--test_samples", type=int, default=10)
    a("--eval_batch_size", type=int, default=16)
    a("--output_dir", type=str, default="ddpm-model-64")
    a("--ddpm_num_steps", type=int, default=1000)
    a("--ddpm_num_inference_steps", type=int, default=250)
    a("--ddpm_beta_schedule", type=str, default="linear")
    a("--prediction_type", type=str, default="epsilon", choices=["epsilon", "sample"])
    a("--use_ema", action="store_true")
    a("--seed", type=int, default=0)
    # accepted for command-line compatibility with scripts/sampling/*.sh; training-only, ignored here
    for flag, typ in (("train_batch_size", int), ("num_epochs", int), ("gradient_accumulation_steps", int),
                      ("learning_rate", float), ("lr_warmup_steps", int), ("lr_scheduler", str),
                      ("mixed_precision", str), ("dataloader_num_workers", int), ("save_images_epochs", int),
                      ("save_model_epochs", int), ("checkpointing_steps", int), ("logger", str),
                      ("logging_dir", str), ("train_data_dir", str), ("dataset_config_name", str),
                      ("model_config_name_or_path", str), ("cache_dir", str), ("resume_from_checkpoint", str),
                      ("adam_beta1", float), ("adam_beta2", float), ("adam_weight_decay", float),
                      ("adam_epsilon", float), ("ema_inv_gamma", float), ("ema_power", float),
                      ("ema_max_decay", float), ("checkpoints_total_limit", int), ("local_rank", int)):
        a(f"--{flag}", type=typ, default=None, help=argparse.SUPPRESS)
    for flag in ("random_flip", "center_crop", "overwrite_output_dir", "push_to_hub",
                 "enable_xformers_memory_efficient_attention"):
        a(f"--{flag}", action="store_true", help=argparse.SUPPRESS)
    g = p.add_argument_group("MI355X build additions")
    g.add_argument("--full_batches", action="store_true")
    g.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    g.add_argument("--root", default=".")
    return p


def build_unet(resolution, dtype, seed):
    from .unet import UNet2DModel
    if resolution not in _LEVELS:
        raise ValueError(f"Unsupported resolution: {resolution}")
    boc, attn = _LEVELS[resolution]
    n = len(boc)
    return UNet2DModel(sample_size=resolution, in_channels=3, out_channels=3, layers_per_block=2,
                       block_out_channels=boc,
                       down_block_types=tuple("AttnDownBlock2D" if i == attn else "DownBlock2D" for i in range(n)),
                       up_block_types=tuple("AttnUpBlock2D" if i == 1 else "UpBlock2D" for i in range(n)),
                       dtype=dtype, seed=seed)


def main(argv=None):
    args = build_parser().parse_args(argv)
    from . import _lib
    from .parallel import gather_images, init_from_env, shard_range
    from .sampler import export_u8
    from .schedulers import DDIMScheduler
    from .unet import UNet2DModel

    rank, world, local = init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("ddim_diffusers.py (MI355X build): no GPU visible; this path has no CPU fallback")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    _lib.load()
    say = print if rank == 0 else (lambda *a, **k: None)
    os.chdir(args.root)
    if args.train_or_test != "test":
        raise SystemExit("training is outside the scope of the MI355X sampling build; use --train_or_test=test")
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)

    out_dir = os.path.join("results_gaussianBN", args.output_dir + ("_ema" if args.use_ema else ""))   # :286-290
    if rank == 0:
        os.makedirs(os.path.join(out_dir, "images"), exist_ok=True)
        os.makedirs(os.path.join(out_dir, "seqs"), exist_ok=True)

    sdir = os.path.join(out_dir, "scheduler")
    if os.path.exists(os.path.join(sdir, "scheduler_config.json")):
        scheduler = DDIMScheduler.from_pretrained(sdir)
    else:
        scheduler = DDIMScheduler(num_train_timesteps=args.ddpm_num_steps, beta_schedule=args.ddpm_beta_schedule,
                                  prediction_type=args.prediction_type)
    scheduler.set_timesteps(args.ddpm_num_inference_steps)
    udir = os.path.join(out_dir, "unet")
    if os.path.exists(os.path.join(udir, "config.json")):
        model = UNet2DModel.from_pretrained(udir, use_safetensors=True, dtype=args.dtype)
    else:
        say(f"[bndm] {udir} not found: sampling from seeded random-init weights")
        model = build_unet(args.resolution, args.dtype, args.seed)
    model = model.to(device).eval()

    path = (f"./results_gaussianBN/{args.dataset_name}_gaussian_linear_outc3_seed0/"
            f"{args.dataset_name}_iadb_gwn_steps250")
    picks = None if args.full_batches else REPLICABILITY_BATCHES.get(args.dataset_name)
    cnt = 0
    for i in range(args.test_samples // args.eval_batch_size):
        if picks is not None and i not in picks:
            continue
        B = args.eval_batch_size
        cached = os.path.join(path, "noise", f"noise_batch{B}_idx{i:05d}.npz")
        if os.path.exists(cached):
            noise = torch.from_numpy(np.load(cached)["noise"]).float()
        else:
            if not args.full_batches:
                say(f"[bndm] {cached} not found: drawing white noise from the numpy stream")
            noise = torch.from_numpy(np.random.randn(B, 3, args.resolution, args.resolution)).float()
        if not args.full_batches:
            noise = noise[0:1]                                            # replicability (:669)
        B = noise.shape[0]
        b0, bc = shard_range(B, rank, world)
        x = noise[b0:b0 + bc].to(device)
        if bc > 0:
            x = scheduler.sample(model, x)                                # loop of :674-681 inside the engine
            u8 = export_u8(x, "round")                                    # (x/2+.5).clamp*255 .round() (:687-688)
        else:
            u8 = torch.empty((0, args.resolution, args.resolution, 3), dtype=torch.uint8, device=device)
        counts = [shard_range(B, r, world)[1] for r in range(world)]
        imgs = gather_images(u8, counts, dst=0)
        if rank == 0:
            from PIL import Image
            for image in imgs.cpu().numpy():
                cnt += 1
                Image.fromarray(image).save(os.path.join(out_dir, "images", f"ddim_img{cnt:05d}.png"))
    say("Done.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
