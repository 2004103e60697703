"""Conditional / super-resolution sampling branch of iadb_bn.py (:566-682): 32 -> 128 px.

x_c = bilinear 4x down + 4x up of the target (align_corners=True, :624-626), x0 = white noise through
``get_noise_v2(..., inplace=True)`` (:630-633), ``sample_iadb_conditional`` with a 6-channel-input UNet
(:635), SSIM / PSNR / L1 / L2 against the target (``piq`` is not available offline; bndm_amd/metrics.py restates
its defaults).  Test images are read from ``./data/<dataset>_test/*/*.png|jpg`` with PIL; without that folder a
seeded synthetic target is used so that the path can be exercised.
"""
from __future__ import annotations

import glob
import os

import numpy as np
import torch

PAPER_PICKS = (74, 104, 278, 389)          # iadb_bn.py:619-622 (1-based indices shown in the paper)


def _load_targets(opt, limit):
    files = sorted(glob.glob(os.path.join("data", f"{opt.dataset}_test", "*", "*.png")) +
                   glob.glob(os.path.join("data", f"{opt.dataset}_test", "*", "*.jpg")))
    if not files:
        rs = np.random.RandomState(opt.seed + 17)
        yy, xx = np.mgrid[0:opt.res, 0:opt.res] / opt.res
        imgs = []
        for i in range(limit):
            f = rs.uniform(1, 4, size=3)
            img = np.stack([0.5 + 0.5 * np.sin(2 * np.pi * (f[c] * xx + (c + i) * 0.1) + yy * 3) for c in range(3)], 0)
            imgs.append(img.astype(np.float32))
        return imgs, False
    from PIL import Image
    out = []
    for f in files[:limit]:
        im = Image.open(f).convert("RGB")
        s = opt.res / min(im.size)
        im = im.resize((max(opt.res, round(im.size[0] * s)), max(opt.res, round(im.size[1] * s))), Image.BILINEAR)
        l, t = (im.size[0] - opt.res) // 2, (im.size[1] - opt.res) // 2
        im = im.crop((l, t, l + opt.res, t + opt.res))
        out.append(np.asarray(im, dtype=np.float32).transpose(2, 0, 1) / 255.0)
    return out, True


def run_conditional(opt, device, rank, world):
    from .bluenoise import get_noise_v2
    from .cli_iadb import NOISE_TAG, output_folder
    from .sampler import export_u8, get_model, sample_iadb_conditional
    from .schedules import get_scheduler_gamma
    from .synth import load_or_make_factor

    say = print if rank == 0 else (lambda *a, **k: None)
    say("===> Start conditional sampling / superres")
    kind = "red" if opt.noise_type == "gaussianRN" else "blue"
    fpath = f"./bluenoise/cov_gaussian{'RN' if kind == 'red' else 'BN'}_L_res64_d3.npz"
    cov_mat_L = torch.from_numpy(load_or_make_factor(fpath, kind)).to(device)
    if opt.noise_type not in ("gaussianBN", "gaussianRN"):
        opt.out_channel = 3
    out_dir = output_folder(opt)
    tag = NOISE_TAG[opt.noise_type]
    folder = f"{opt.dataset}_iadb_{tag}_{opt.conditional_type}_steps{opt.nb_steps}"
    if rank == 0:
        for sub in ("images", "seqs", "lowres", "highres"):
            os.makedirs(os.path.join(out_dir, folder, sub), exist_ok=True)
    if opt.conditional_type != "superres":
        raise NotImplementedError(opt.conditional_type)
    model = get_model(6, opt.out_channel, opt.res, activation=opt.activation, dtype=opt.dtype, seed=opt.seed)
    ckpt = os.path.join(out_dir, "model.ckpt")
    if os.path.exists(ckpt):
        model.load_state_dict(torch.load(ckpt, map_location="cpu"))
    else:
        say(f"[bndm] {ckpt} not found: sampling from seeded random-init weights")
    model = model.to(device).eval()
    sp = np.array([opt.scheduler_param, opt.scheduler_param_s, opt.scheduler_param_e]).astype(np.float32)
    scheduler_params = torch.from_numpy(sp).float().to(device)

    targets, real = _load_targets(opt, max(PAPER_PICKS) if not opt.full_batches else opt.test_samples)
    picks = range(1, len(targets) + 1) if (opt.full_batches or not real) else PAPER_PICKS
    if not real:
        say(f"[bndm] ./data/{opt.dataset}_test not found: using seeded synthetic targets")
        picks = range(1, min(len(targets), 4) + 1)
    psnr_sum, ssim_sum, l1_sum, l2_sum, cnt = 0.0, 0.0, 0.0, 0.0, 0
    from .metrics import ssim
    from PIL import Image
    produced = 0
    for idx in picks:
        if idx > len(targets):
            continue
        # every rank consumes the draw of every pick (the reference's single numpy stream, iadb_bn.py:631), so the
        # images do not depend on the number of ranks; a rank keeps the picks it owns
        x0_np = np.random.randn(1, 3, opt.res, opt.res)
        order = produced
        produced += 1
        cnt_name = produced                                               # running cnt of iadb_bn.py:662: images produced
        if order % world != rank:
            continue
        x1 = torch.from_numpy(targets[idx - 1])[None].to(device) * 2 - 1
        lo = opt.res // 4
        x_c = torch.nn.functional.interpolate(x1, size=(lo, lo), mode="bilinear", align_corners=True)
        x_c = torch.nn.functional.interpolate(x_c, size=(opt.res, opt.res), mode="bilinear", align_corners=True)
        x0 = torch.from_numpy(x0_np).float().to(device)
        t = torch.full((1,), opt.nb_steps, device=device)
        gamma_t = get_scheduler_gamma(t.float(), opt.scheduler_gamma, scheduler_params, opt.nb_steps)
        x0, _, _ = get_noise_v2(device, x0, cov_mat_L, gamma_t, t, noise_type=opt.noise_type, train_or_test="test",
                                inplace=True)
        sample, sample_all = sample_iadb_conditional(model, x0, x_c, opt.nb_steps, opt.scheduler_gamma,
                                                     scheduler_params, opt.out_channel, opt.noise_type, "test",
                                                     scheduler_alpha=opt.scheduler_alpha,
                                                     alpha_param=opt.scheduler_param)
        rec = torch.clamp((sample + 1) / 2, 0, 1)
        ref = (x1 + 1) / 2
        mse = torch.mean((rec - ref) ** 2).item()
        psnr_sum += 10 * np.log10(1.0 / max(mse, 1e-12))
        ssim_sum += float(ssim(rec, ref, data_range=1.0)[0])                      # iadb_bn.py:639
        l2_sum += torch.sum((sample - x1) ** 2).item()
        l1_sum += torch.sum(torch.abs(sample - x1)).item()
        cnt += 1
        u8 = export_u8(sample, "trunc")[0].cpu().numpy()
        Image.fromarray(u8).save(os.path.join(out_dir, folder, "images", f"image_{tag}_{cnt_name:05d}.png"))
        if opt.noise_type == "gaussian":
            Image.fromarray(export_u8(x1, "trunc")[0].cpu().numpy()).save(
                os.path.join(out_dir, folder, "highres", f"highres_{tag}_{cnt_name:05d}.png"))
            Image.fromarray(export_u8(x_c, "trunc")[0].cpu().numpy()).save(
                os.path.join(out_dir, folder, "lowres", f"lowres_{tag}_{cnt_name:05d}.png"))
    # the averages cover every image of every rank (iadb_bn.py:681 prints one line for the whole test set)
    sums = torch.tensor([ssim_sum, psnr_sum, l1_sum, l2_sum, float(cnt)], dtype=torch.float64, device=device)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(sums)
    tot = int(sums[4].item())
    if tot and rank == 0:
        a = (sums[:4] / sums[4]).tolist()
        print(f"conditional metrics over {tot} images: ssim {a[0]:.4f}, psnr {a[1]:.4f}, l1 {a[2]:.2f}, l2 {a[3]:.2f}")
    return 0
