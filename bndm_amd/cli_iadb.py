"""``python iadb_bn.py --train_or_test=test ...`` on MI355X: the sampling half of the reference's
iadb_bn.py (flags :29-67, output tree :481-499/:689-707, unconditional branch :686-821, conditional /
super-resolution branch :566-682).

Same command lines as scripts/sampling/*.sh.  Differences, all additive:
  * ``--full_batches`` lifts the reference's "replicability" clamps (hard-coded batch ids
    :744-753, pre-saved noise + single sample :763-766) so whole batches are sampled, x0 goes through
    ``get_noise_v2(..., inplace=True)`` (the call commented out at :775) and ``--save_noise`` writes
    the ``noise_batch<B>_idx<i>.npz`` cache (:783);
  * without ``./bluenoise/cov_gaussian*_L_res64_d3.npz`` / ``model.ckpt`` (neither ships with the
    reference, README.md:33-36) a synthetic blue factor / seeded random weights are used, loudly;
  * launched under ``torch.distributed.run`` the batch is sharded over the ranks (one RCCL gather of
    the uint8 images per batch) instead of ``torch.nn.DataParallel`` (:716);
  * training is out of scope for this build and exits with a message.
"""
from __future__ import annotations

import argparse
import glob
import os
import random
import sys
import time

import numpy as np
import torch

FLAGS = [
    # (name, type, default)
    ("dataset", str, "celeba_small"), ("noise_type", str, "gaussian"), ("optimizer_type", str, "adamw"),
    ("epochs", int, 20), ("batch_size", int, 64), ("res", int, 64), ("train_or_test", str, "train"),
    ("checkpoint", str, None), ("seed", int, 0), ("nb_steps", int, 1000), ("scheduler_alpha", str, "linear"),
    ("scheduler_gamma", str, "linear"), ("scheduler_param", float, 0.02), ("scheduler_param_s", float, 0),
    ("scheduler_param_e", float, 3), ("blue_noise_blur", float, None), ("activation", str, "silu"),
    ("early_stopping_step", int, 50), ("split_step", int, 900), ("lr", float, 1e-4), ("mode_index", int, 1),
    ("reg_weight", float, 1), ("alpha_min", float, 0.0), ("grad_clip", float, None), ("deterministic", int, 1),
    ("conditional_type", str, "superres"), ("fine_tune_mode_index", int, 0), ("skip", int, 1),
    ("test_samples", int, 10), ("out_channel", int, 6),
]
SWITCHES = ["resume_training", "optimize_scheduler_param", "remap", "is_conditional"]

REPLICABILITY_BATCHES = {            # iadb_bn.py:744-753
    "cat_res64": [4], "cat_res128": [52], "celeba_res64": [37], "celeba_res128": [10],
    "church_res64": [4, 23, 32, 36],
}
NOISE_TAG = {"gaussianBN": "gwn2gbn", "gaussian": "gwn", "gaussianRN": "gwn2grn", "GBN": "gbn"}


def build_parser():
    p = argparse.ArgumentParser(description="bndm IADB sampling on MI355X (drop-in for iadb_bn.py)")
    for name, typ, default in FLAGS:
        p.add_argument(f"--{name}", type=typ, default=default)
    for name in SWITCHES:
        p.add_argument(f"--{name}", action="store_true")
    g = p.add_argument_group("MI355X build additions")
    g.add_argument("--full_batches", action="store_true", help="sample whole batches (no replicability clamps)")
    g.add_argument("--save_noise", action="store_true", help="write noise/noise_batch<B>_idx<i>.npz")
    g.add_argument("--dtype", default="f16", choices=["f16", "bf16"], help="UNet storage / MFMA input type")
    g.add_argument("--root", default=".", help="directory holding bluenoise/, results_gaussianBN/, data/")
    return p


def output_folder(opt):
    """iadb_bn.py:481-496."""
    outer = f"results_gaussianBN_{opt.conditional_type}" if opt.is_conditional else "results_gaussianBN"
    if opt.scheduler_gamma == "linear" or opt.optimize_scheduler_param:
        tail = f"{opt.dataset}_{opt.noise_type}_{opt.scheduler_gamma}_outc{opt.out_channel}_seed{opt.seed}"
    else:
        mid = f"{opt.scheduler_param}_{opt.scheduler_param_s}_{opt.scheduler_param_e}"
        remap = "_remap" if opt.remap else ""
        tail = (f"{opt.dataset}_{opt.noise_type}_{opt.scheduler_gamma}_{mid}_outc{opt.out_channel}"
                f"{remap}_seed{opt.seed}")
    return os.path.join(outer, tail)


def _save_png(arr_u8, path):
    from PIL import Image
    Image.fromarray(arr_u8).save(path)


def _minmax_u8(t):
    t = (t - t.min()) / (t.max() - t.min())
    return (t.permute(1, 2, 0).detach().cpu().numpy() * 255).astype(np.uint8)


def main(argv=None):
    opt = build_parser().parse_args(argv)
    from . import _lib
    from .bluenoise import get_noise_v2
    from .parallel import gather_images, init_from_env, shard_range
    from .sampler import export_u8, get_model, sample_iadb
    from .schedules import get_scheduler_gamma
    from .synth import load_or_make_factor

    torch.cuda.manual_seed(opt.seed)
    torch.manual_seed(opt.seed)
    np.random.seed(opt.seed)
    random.seed(opt.seed)
    rank, world, local = init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("iadb_bn.py (MI355X build): no GPU visible; this path has no CPU fallback")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    _lib.load()
    say = print if rank == 0 else (lambda *a, **k: None)
    os.chdir(opt.root)

    if opt.train_or_test != "test":
        raise SystemExit("training is outside the scope of the MI355X sampling build (SURVEY.md section 2); "
                         "use --train_or_test=test")
    if opt.is_conditional:
        from .cli_superres import run_conditional
        return run_conditional(opt, device, rank, world)

    kind = "red" if opt.noise_type == "gaussianRN" else "blue"
    fpath = f"./bluenoise/cov_gaussian{'RN' if kind == 'red' else 'BN'}_L_res64_d3.npz"
    if not os.path.exists(fpath):
        say(f"[bndm] {fpath} not found (not shipped with the reference): using the synthetic {kind} factor")
    cov_mat_L = torch.from_numpy(load_or_make_factor(fpath, kind)).to(device)

    if opt.noise_type not in ("gaussianBN", "gaussianRN"):
        opt.out_channel = 3                                              # iadb_bn.py:476-479
    out_dir = output_folder(opt)
    if opt.noise_type not in NOISE_TAG:
        raise NotImplementedError
    tag = NOISE_TAG[opt.noise_type]
    folder = f"{opt.dataset}_iadb_{tag}_steps{opt.nb_steps}"
    if rank == 0:
        for sub in ("images", "seqs", "noise"):
            os.makedirs(os.path.join(out_dir, folder, sub), exist_ok=True)

    say("===> Start unconditional sampling")
    model = get_model(3, opt.out_channel, opt.res, activation=opt.activation, dtype=opt.dtype, seed=opt.seed)
    ckpt = os.path.join(out_dir, "model.ckpt")
    if os.path.exists(ckpt):
        model.load_state_dict(torch.load(ckpt, map_location="cpu"))
    else:
        say(f"[bndm] {ckpt} not found: sampling from seeded random-init weights (images are noise-like)")
    model = model.to(device).eval()

    total = opt.test_samples
    num_batch = total // opt.batch_size if total % opt.batch_size == 0 else total // opt.batch_size + 1
    last_bs = opt.batch_size if total % opt.batch_size == 0 else total - (num_batch - 1) * opt.batch_size
    say("num_batch:", num_batch)
    if opt.optimize_scheduler_param:
        sp = np.loadtxt(os.path.join(out_dir, "scheduler_params.txt"))
    else:
        sp = np.array([opt.scheduler_param, opt.scheduler_param_s, opt.scheduler_param_e]).astype(np.float32)
    scheduler_params = torch.from_numpy(np.asarray(sp)).float().to(device)

    cnt = 0
    fwd_times, noise_times = [], []
    picks = None if opt.full_batches else REPLICABILITY_BATCHES.get(opt.dataset)
    for i in range(num_batch):
        if picks is not None and i not in picks:
            continue
        B = last_bs if i == num_batch - 1 else opt.batch_size
        cur_bs = B                                                       # cur_batch_size of iadb_bn.py:757-759
        # the global numpy stream is consumed exactly as at iadb_bn.py:761 (rank-independent)
        x0 = torch.from_numpy(np.random.randn(B, 3, opt.res, opt.res)).float()
        if not opt.full_batches:
            cached = (f"./results_gaussianBN/{opt.dataset}_gaussian_linear_outc3_seed0/"
                      f"{opt.dataset}_iadb_gwn_steps250/noise/noise_batch{opt.batch_size}_idx{i:05d}.npz")
            if os.path.exists(cached):
                x0 = torch.from_numpy(np.load(cached)["noise"]).float()
            else:
                say(f"[bndm] {cached} not found: using this batch's np.random draw")
            x0 = x0[0:1]                                                 # replicability: one sample (:766)
            B = 1
        b0, bc = shard_range(B, rank, world)
        x0 = x0.to(device)
        xg = x0                                                          # global batch (every rank holds it)
        t = torch.full((B,), opt.nb_steps, device=device)
        gamma_t = get_scheduler_gamma(t.float(), opt.scheduler_gamma, scheduler_params, opt.nb_steps)
        t0 = time.time()
        if opt.full_batches:
            x0, _, _ = get_noise_v2(device, x0, cov_mat_L, gamma_t, t, noise_type=opt.noise_type,
                                    train_or_test="test", inplace=True, batch_range=(b0, bc))
        else:
            x0 = x0[b0:b0 + bc]
        torch.cuda.synchronize()
        noise_times.append(time.time() - t0)
        if opt.save_noise and opt.full_batches and rank == 0:
            x_full = x0 if world == 1 else get_noise_v2(device, xg, cov_mat_L, gamma_t, t, noise_type=opt.noise_type,
                                                        train_or_test="test", inplace=True)[0]
            np.savez_compressed(os.path.join(out_dir, folder, "noise", f"noise_batch{B}_idx{i:05d}.npz"),
                                noise=x_full.cpu().numpy())
        if bc > 0:
            sample, sample_all, ft = sample_iadb(model, x0, opt.nb_steps, opt.scheduler_gamma, scheduler_params,
                                                 opt.out_channel, opt.noise_type, "test",
                                                 scheduler_alpha=opt.scheduler_alpha, log_freq=25,
                                                 alpha_param=opt.scheduler_param)      # iadb_bn.py:115,131
            fwd_times.append(ft)
            u8 = export_u8(sample, "trunc")
        else:
            sample_all, u8 = [], torch.empty((0, opt.res, opt.res, 3), dtype=torch.uint8, device=device)
        counts = [shard_range(B, r, world)[1] for r in range(world)]
        imgs = gather_images(u8, counts, dst=0)
        if rank == 0:
            for j, snap in enumerate(sample_all):                        # seqs of this rank's first sample
                frame = snap[0]
                if j == len(sample_all) - 1:
                    arr = (torch.clamp((frame + 1) / 2.0, 0.0, 1.0).permute(1, 2, 0).cpu().numpy() * 255).astype(np.uint8)
                else:
                    arr = _minmax_u8(frame)
                _save_png(arr, os.path.join(out_dir, folder, "seqs",
                                            f"{tag}_img{cnt:05d}_step{int((j * 100) / 1000 * opt.nb_steps)}.png"))
            imgs = imgs.cpu().numpy()
            # iadb_bn.py:808-816: cnt advances once per sample of the batch (batch_size of them), also in
            # replicability mode, where only sample 0 is kept -- 00001.png, 00501.png, ... at batch_size 500
            for j in range(cur_bs):
                cnt += 1
                if j < B:
                    _save_png(imgs[j], os.path.join(out_dir, folder, "images", f"{cnt:05d}.png"))
    say("np.mean(inference_times) per UNet forward", float(np.mean(fwd_times)) if fwd_times else float("nan"))
    say("np.mean(noise_gen_times)", float(np.mean(noise_times[1:])) if len(noise_times) > 1 else float("nan"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
