#!/usr/bin/env python
"""Headline benchmark: images/sec of the bndm sampling hot path on MI355X.

Workload (BASELINE.json configs[1]): cat_res64 IADB, batch 64 per GPU, 250 Euler steps,
UNet 3 -> 6 channels with the sigmoid(tau=1000, 0, 3) white<->blue gamma schedule, x0 from the tiled
blue-noise generator (64x64 tiles of the 4096x4096 factor).  One "step" of this benchmark is one
full pass of that path over one batch: white draw -> get_noise_v2 (L.z) -> 250 x (UNet forward +
Euler update) -> uint8 export (-> one RCCL gather to rank 0 when N > 1).  Synthetic L (blue
Cholesky factor), seeded random-init weights of the reference architecture; inputs are resident in
HBM when the timed region starts.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 3 --warmup 1

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_TFLOPS = 2500.0       # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PMC_TRAFFIC_PER_LAUNCH = 131.4e6  # B, average over conv_tap9<TH=16> launches (profiles/r01_pmc_traffic.txt)


def cpu_baseline(nb_steps, seed=0):
    """Reference op sequence on the host cores (oracle = fp32 restatement; the reference's Python
    cannot travel to the GPU box): dense torch.matmul noise transform + 3 timed UNet/Euler steps at
    B=4 after one warm-up, extrapolated to nb_steps.  A reported baseline, not the target."""
    from oracle import noise_oracle as NO
    from oracle import sampler_oracle as SO
    from oracle import unet_oracle as UO
    from bndm_amd.synth import formula_factor
    B = 4
    cfg = UO.make_config(64, 3, 6)
    sd = UO.init_params(cfg, seed=seed)
    L = formula_factor()
    rs = np.random.RandomState(seed)
    z = rs.standard_normal((B, 3, 64, 64)).astype(np.float32)
    t0 = time.perf_counter()
    x0, _, _ = NO.get_noise_v2(z, L, np.ones(B, np.float32), "gaussianBN", "test")
    t_noise = time.perf_counter() - t0
    x = torch.from_numpy(np.ascontiguousarray(x0))
    params = torch.tensor([1000.0, 0.0, 3.0])
    model = UO.OracleUNet(cfg, sd)

    def one_step(x, t):
        tt = torch.full((B,), t, dtype=torch.int64)
        a1 = SO.alpha_schedule((tt + 1).float(), nb_steps)
        a0 = SO.alpha_schedule(tt.float(), nb_steps)
        g1 = SO.gamma_schedule((tt + 1).float(), nb_steps, "sigmoid", params)
        g0 = SO.gamma_schedule(tt.float(), nb_steps, "sigmoid", params)
        d = model(x, a1)[0]
        return x + (a1 - a0).view(-1, 1, 1, 1) * d[:, :3] + (g1 - g0).view(-1, 1, 1, 1) * d[:, 3:]

    # thread count: all host cores oversubscribe MKL-DNN on big boxes (256 threads: >50x slower than
    # 32); calibrate on one step per candidate and keep the fastest
    ncpu = os.cpu_count() or 1
    best = None
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        one_step(x, nb_steps - 1)                    # warm-up at this thread count
        t0 = time.perf_counter()
        one_step(x, nb_steps - 1)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, nt)
        if dt > 20:
            break
    torch.set_num_threads(best[1])
    x = one_step(x, nb_steps - 1)                    # warm-up
    n_timed = 3
    t0 = time.perf_counter()
    for k in range(n_timed):
        x = one_step(x, nb_steps - 2 - k)
    t_step = (time.perf_counter() - t0) / n_timed
    total = t_noise + nb_steps * t_step
    return {
        "value": B / total, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"B={B}: dense L.z noise ({t_noise*1e3:.0f} ms) + {n_timed} timed fp32 UNet+Euler steps "
                  f"({t_step*1e3:.0f} ms/step) after 1 warm-up, extrapolated to {nb_steps} steps",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per pass")
    ap.add_argument("--nb_steps", type=int, default=250, help="denoising steps per image")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-only", action="store_true", help="skip the timed passes; only the per-kernel profile")
    args = ap.parse_args()

    from bndm_amd import _lib
    from bndm_amd.bluenoise import get_noise_v2
    from bndm_amd.parallel import barrier, gather_images, init_from_env, max_over_ranks
    from bndm_amd.sampler import export_u8, get_model, sample_iadb
    from bndm_amd.schedules import get_scheduler_gamma
    from bndm_amd.synth import blue_noise_factor

    rank, world, local = init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    _lib.load()

    B, N = args.batch, args.nb_steps
    torch.manual_seed(1234 + rank)
    L = torch.from_numpy(blue_noise_factor("blue")).to(dev)
    model = get_model(3, 6, 64, dtype=args.dtype, seed=0).to(dev).eval()
    params = torch.tensor([1000.0, 0.0, 3.0], device=dev)
    gamma_T = get_scheduler_gamma(torch.full((B,), float(N), device=dev), "sigmoid", params, N)
    t_full = torch.full((B,), N, device=dev)

    def one_pass():
        z = torch.randn(B, 3, 64, 64, device=dev)                                      # on-device Philox
        x0, _, _ = get_noise_v2(dev, z, L, gamma_T, t_full, noise_type="gaussianBN", train_or_test="test",
                                inplace=True)
        x = sample_iadb(model, x0, N, "sigmoid", params, 6, "gaussianBN", "train")
        u8 = export_u8(x, "trunc")
        return gather_images(u8, dst=0)

    elapsed = float("nan")
    if not args.profile_only:
        for _ in range(args.warmup):
            one_pass()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = one_pass()
        barrier()
        torch.cuda.synchronize()
        elapsed = max_over_ranks(time.perf_counter() - t0, device=dev)

    # ---- dominant kernel (conv_igemm) timed with HIP events on the launch stream ---------------------
    roof = None
    if rank == 0:
        lib = _lib.load()
        prof = _lib.UNetProfile()
        core = model
        h = core._ensure_engine(B, 64, dev)
        xs = torch.randn(B, 3, 64, 64, device=dev)
        ts = torch.full((B,), 0.5, device=dev)
        od = torch.empty(B, 6, 64, 64, device=dev)
        rc = lib.bndm_unet_profile(h, C.c_void_p(xs.data_ptr()), C.c_void_p(ts.data_ptr()), C.c_void_p(od.data_ptr()),
                                   B, 3, C.byref(prof), _lib.current_stream_ptr())
        _lib.check(rc, "bndm_unet_profile")
        achieved = prof.dom_flops / (prof.ms_dom * 1e-3) / 1e12
        conv_all = prof.conv_flops / (prof.ms_conv * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "conv_tap9<TH=16> (GroupNorm+SiLU fused 3x3 conv, 256-pixel tiles)",
                "achieved": round(achieved, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_F16_TFLOPS, 4),
                # HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE
                # doubled per MI355X_MICROARCH.md + WRITE_SIZE), see profiles/r01_pmc_traffic.txt
                "traffic": PMC_TRAFFIC_PER_LAUNCH,
                "launches_per_forward": prof.dom_launches,
                "avg_launch_us": round(prof.ms_dom / max(prof.dom_launches, 1) * 1e3, 2),
                "algorithmic_flop_per_launch": prof.dom_flops / max(prof.dom_launches, 1),
                "algorithmic_bytes_per_launch": prof.dom_bytes / max(prof.dom_launches, 1),
                "all_conv_kernels_tflops": round(conv_all, 1), "ms_per_forward_conv": round(prof.ms_conv, 3),
                "ms_per_forward_total": round(prof.ms_total, 3), "launches_total": prof.launches}

    # ---- per-stage figures: blue-noise transform vs its HBM / fp32-MFMA rooflines -------------
    stages = None
    if rank == 0:
        def time_noise(nb, reps=20):
            zz = torch.randn(nb, 3, 64, 64, device=dev)
            aa = torch.ones(nb, device=dev)
            tt = torch.full((nb,), N, device=dev)
            for _ in range(3):
                get_noise_v2(dev, zz, L, aa, tt, noise_type="GBN", train_or_test="test", inplace=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                get_noise_v2(dev, zz, L, aa, tt, noise_type="GBN", train_or_test="test", inplace=True)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3            # us per call (gemm + finish kernels)
        TRI = 4096 * 4097 // 2
        us_small, us_batch = time_noise(2), time_noise(B)
        n_small, n_batch = 2 * 3, B * 3
        stages = {
            "bluenoise_B2_us": round(us_small, 1),
            "bluenoise_B2_GBps_of_L": round(4 * TRI / (us_small * 1e-6) / 1e9, 1),      # HBM-bound regime (n = 6 columns)
            "bluenoise_B%d_us" % B: round(us_batch, 1),
            "bluenoise_B%d_fp32_TFLOPs" % B: round(2.0 * TRI * n_batch / (us_batch * 1e-6) / 1e12, 2),   # fp32-MFMA-bound regime
            "hbm_peak_GBps": 8000, "fp32_mfma_peak_TFLOPs": 157.3,
            "unet_ms_per_denoising_step": roof["ms_per_forward_total"] if roof else None,
        }

    if rank == 0:
        imgs = args.gpus * B * args.steps
        line = {
            "metric": "images/sec, IADB 64x64 UNet, 250 steps, tiled blue noise",
            "value": round(imgs / elapsed, 3), "unit": "images/sec", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"cat_res64 IADB, batch={B}/GPU, {N} steps, gaussianBN sigmoid(1000,0,3), "
                                   f"UNet 3->6, tiled Gaussian blue noise (64^2 tiles from 4096^2 L)",
                       "global_batch": B * args.gpus, "nb_steps": N, "parallelism": f"batch-shard x{args.gpus}"},
            "roofline": roof,
            "stages": stages,
            # timed on rank 0 of the single-GPU run only (the host cores are shared by the ranks otherwise)
            "cpu_baseline": None if (args.no_cpu_baseline or args.gpus > 1) else cpu_baseline(N),
        }
        print(json.dumps(line), flush=True)
    barrier()


if __name__ == "__main__":
    main()
