#!/usr/bin/env python
"""Headline benchmark: images/sec of the bndm sampling hot path on MI355X.

Default workload (BASELINE.json configs[1], ``--config c2``): cat_res64 IADB, batch 64 per GPU, 250 Euler
steps, UNet 3 -> 6 channels with the sigmoid(tau=1000, 0, 3) white<->blue gamma schedule, x0 from the tiled
blue-noise generator (64x64 tiles of the 4096x4096 factor).  One "step" of this benchmark is one full pass of
that path over one batch: white draw -> get_noise_v2 (L.z) -> 250 x (UNet forward + Euler update) -> uint8 export
(-> one RCCL gather to rank 0 when N > 1).  Synthetic L (blue Cholesky factor), seeded random-init weights of
the reference architecture; inputs are resident in HBM when the timed region starts.

The other BASELINE.json configurations emit the same JSON shape (``config.workload`` names them):
    --config c3   church_res64 DDIM (ddim_diffusers.py), 100 steps, UNet 3 -> 3, batch 64 per GPU
    --config c4   celeba_res128 IADB, 250 steps, UNet 3 -> 6, sigmoid(0.2, 0, 3); batch 32 per GPU (= 256 / 8 GPUs)
    --config c5   latent_iadb_cat_res512: latent UNet 4 -> 8, 250 steps, + VAE decode to 512x512; batch 8 per GPU

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus 8 --steps 3 --warmup 1          # spawns its own 8 ranks (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 3 --warmup 1

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_TFLOPS = 2500.0       # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # written by tools/pmc_traffic.py


def host_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"nproc": os.cpu_count() or 1, "cpu_model": model}


def cpu_baseline(seed=0, budget_s=45.0):
    """BASELINE.json configs[0] on the host cores: cat_res64 IADB, B=4, white noise ('gaussian'), UNet 3 -> 3,
    250 steps, run IN FULL with the oracle (fp32 restatement of the reference's op sequence: the reference's Python
    cannot travel to the GPU box).  If the host is too slow to finish 250 steps inside `budget_s`, the loop stops
    there and the remainder is extrapolated -- the sample string says which.  A reported baseline, not the target."""
    import numpy as np
    import torch
    from oracle import sampler_oracle as SO
    from oracle import unet_oracle as UO
    B, N = 4, 250
    cfg = UO.make_config(64, 3, 3)
    sd = UO.init_params(cfg, seed=seed)
    rs = np.random.RandomState(seed)
    x = torch.from_numpy(rs.standard_normal((B, 3, 64, 64)).astype(np.float32))    # white start (iadb_bn.py:761)
    model = UO.OracleUNet(cfg, sd)

    def one_step(x, t):
        tt = torch.full((B,), t, dtype=torch.int64)
        a1 = SO.alpha_schedule((tt + 1).float(), N)
        a0 = SO.alpha_schedule(tt.float(), N)
        d = model(x, a1)[0]
        return x + (a1 - a0).view(-1, 1, 1, 1) * d

    # thread count: all host cores oversubscribe MKL-DNN on big boxes (256 threads: >50x slower than 32);
    # calibrate on one step per candidate and keep the fastest
    hi = host_info()
    best = None
    for nt in sorted({min(hi["nproc"], c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        one_step(x, N - 1)                           # warm-up at this thread count
        t0 = time.perf_counter()
        one_step(x, N - 1)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, nt)
        if dt > 10:
            break
    torch.set_num_threads(best[1])
    t0 = time.perf_counter()
    done = 0
    for s in range(N):
        x = one_step(x, N - 1 - s)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    total = el * N / done
    how = "run in full" if done == N else f"{done} of {N} steps timed ({el:.1f} s), extrapolated to {N}"
    return {
        "value": B / total, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
        "nproc": hi["nproc"], "cpu_model": hi["cpu_model"], "torch_threads": torch.get_num_threads(),
        "sample": f"BASELINE configs[0]: cat_res64 IADB B={B}, white noise, UNet 3->3 fp32 oracle, {N} steps: {how} "
                  f"({total / N * 1e3:.0f} ms/step on {torch.get_num_threads()} threads)",
    }


def pmc_traffic(lib_path, kernel_tag, config):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes of tools/pmc_traffic.py -- only if
    they were taken from THIS build of the library (sha256 match) on THIS workload (the PMC passes run the default
    configuration, c2); otherwise null (never a stale or foreign constant)."""
    try:
        with open(PMC_TRAFFIC_FILE) as f:
            rec = json.load(f)
        with open(lib_path, "rb") as f:
            sha = hashlib.sha256(f.read()).hexdigest()
        if rec.get("lib_sha256") == sha and rec.get("config", "c2") == config and kernel_tag in rec.get("kernels", {}):
            return rec["kernels"][kernel_tag]["bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU,
    torch.distributed.run on 127.0.0.1) and pass their single JSON line through."""
    if not args.dry_run_cpu:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible -- refusing to report a "
                             f"{args.gpus}-GPU line from fewer devices")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


# -------------------------------------------------------------------------------------------------
# workloads
# -------------------------------------------------------------------------------------------------
def make_workload(name, B, N, dtype, dev, rank=0, world=1):
    """-> dict(one_pass, model, res, cin, cout, images_per_pass, metric, workload, extra_stage)

    c4 on more than one GPU is the named configuration "batch = world x 32 sharded": the 128-px blue-noise branch permutes
    tiles ACROSS the batch (get_noise_recent.py:131-146), so every rank draws the same global batch (one seeded Philox
    stream per pass, as iadb_bn.py:761 draws one global numpy batch) and computes its own contiguous shard of
    get_noise_v2's result with ``batch_range`` -- bit-identical to the 1-GPU result on the global batch
    (tests/test_gpu_noise.py::test_batch_range_sharding_matches_global, at B=256:
    tests/test_gpu_layouts.py::test_c4_global_batch_256_shards_equal_the_one_gpu_result)."""
    import torch
    from bndm_amd.bluenoise import get_noise_v2
    from bndm_amd.sampler import export_u8, get_model, sample_iadb
    from bndm_amd.schedules import get_scheduler_gamma
    from bndm_amd.synth import blue_noise_factor

    if name in ("c2", "c4"):
        res = 64 if name == "c2" else 128
        B = B or (64 if name == "c2" else 32)
        N = N or 250
        tau = 1000.0 if name == "c2" else 0.2
        L = torch.from_numpy(blue_noise_factor("blue")).to(dev)
        model = get_model(3, 6, res, dtype=dtype, seed=0).to(dev).eval()
        params = torch.tensor([tau, 0.0, 3.0], device=dev)
        sharded = name == "c4" and world > 1
        BG = B * world if sharded else B                                              # batch the noise branch sees
        gamma_T = get_scheduler_gamma(torch.full((BG,), float(N), device=dev), "sigmoid", params, N)
        t_full = torch.full((BG,), N, device=dev)
        gen = torch.Generator(device=dev)
        passes = [0]

        def one_pass():
            if sharded:
                from bndm_amd.parallel import shard_range
                gen.manual_seed(977 + passes[0])                                       # the SAME global draw on every rank
                passes[0] += 1
                z = torch.randn(BG, 3, res, res, device=dev, generator=gen)
                x0, _, _ = get_noise_v2(dev, z, L, gamma_T, t_full, noise_type="gaussianBN", train_or_test="test",
                                        inplace=True, l_is_triangular=True, batch_range=shard_range(BG, rank, world))
            else:
                z = torch.randn(B, 3, res, res, device=dev)                           # on-device Philox
                x0, _, _ = get_noise_v2(dev, z, L, gamma_T, t_full, noise_type="gaussianBN", train_or_test="test",
                                        inplace=True, l_is_triangular=True)
            x = sample_iadb(model, x0, N, "sigmoid", params, 6, "gaussianBN", "train")
            return export_u8(x, "trunc")
        ds = "cat_res64" if name == "c2" else "celeba_res128"
        tiles = "64^2 tiles" if res == 64 else "2x2 tiles of 64^2 per image"
        return dict(one_pass=one_pass, model=model, res=res, cin=3, cout=6, B=B, N=N, L=L,
                    metric=f"images/sec, IADB {res}x{res} UNet, {N} steps, tiled blue noise",
                    workload=f"{ds} IADB, batch={B}/GPU, {N} steps, gaussianBN sigmoid({tau:g},0,3), UNet 3->6, "
                             f"tiled Gaussian blue noise ({tiles} from 4096^2 L)" +
                             (f"; global batch {BG} drawn once per pass, tile permutation over the global batch, "
                              f"sharded {B}/GPU by batch_range" if sharded else ""))
    if name == "c3":
        from bndm_amd.schedulers import DDIMScheduler
        B, N = B or 64, N or 100
        model = get_model(3, 3, 64, dtype=dtype, seed=0).to(dev).eval()
        sch = DDIMScheduler(num_train_timesteps=1000, beta_schedule="linear")
        sch.set_timesteps(N)

        def one_pass():
            x = torch.randn(B, 3, 64, 64, device=dev)
            return export_u8(sch.sample(model, x), "round")
        return dict(one_pass=one_pass, model=model, res=64, cin=3, cout=3, B=B, N=N, L=None,
                    metric=f"images/sec, DDIM 64x64 UNet, {N} steps",
                    workload=f"church_res64 DDIM (eps-prediction, eta 0, clip 1), batch={B}/GPU, {N} steps, UNet 3->3, "
                             f"white x0 (the reference's DDIM script has no white->blue schedule, SURVEY 0.5)")
    if name == "c5":
        from bndm_amd.schedulers import IADBScheduler
        from bndm_amd.unet import UNet2DModel
        from bndm_amd.vae import AutoencoderKL, vae_decode
        B, N = B or 8, N or 250
        boc = (128, 128, 256, 256, 512, 512)
        model = UNet2DModel(sample_size=64, in_channels=4, out_channels=8, layers_per_block=2, block_out_channels=boc,
                            down_block_types=tuple("AttnDownBlock2D" if i == 4 else "DownBlock2D" for i in range(6)),
                            up_block_types=tuple("AttnUpBlock2D" if i == 1 else "UpBlock2D" for i in range(6)),
                            dtype=dtype, seed=0).to(dev).eval()
        vae = AutoencoderKL(dtype=dtype).to(dev).eval()
        sch = IADBScheduler(noise_type="gaussianBN", out_channels=8)
        sch.set_timesteps(N)
        L = torch.from_numpy(blue_noise_factor("blue")).to(dev)
        ones = torch.ones(B, device=dev)

        def one_pass():
            z = torch.randn(B, 4, 64, 64, device=dev)
            x0, _, _ = get_noise_v2(dev, z, L, ones, None, noise_type="GBN", train_or_test="test", inplace=True,
                                    l_is_triangular=True)
            lat = sch.sample(model, x0)
            return export_u8(vae_decode(vae, lat), "round")
        return dict(one_pass=one_pass, model=model, res=64, cin=4, cout=8, B=B, N=N, L=L, vae=vae,
                    metric=f"images/sec, latent IADB (64x64x4 latents) {N} steps + VAE decode to 512x512",
                    workload=f"latent_iadb_cat_res512, batch={B}/GPU, {N} steps, latent UNet 4->8, blue-noise x0 in "
                             f"latent space (GBN), AutoencoderKL decode 64^2 -> 512^2")
    raise SystemExit(f"unknown --config {name}")


def short_pass(name, dtype, dev, lib, timed=2):
    """One BASELINE.json configuration at its per-GPU size: 1 warm-up + `timed` passes of the whole path, plus the
    forward's event-timed duration and TFLOP/s (algorithmic 2*MAC of the engine's launch list)."""
    import torch
    from bndm_amd import _lib
    from bndm_amd.unet import engine_ops
    os.environ.pop("BNDM_PROFILE_DUMP", None)    # the per-op dump is the headline workload's, written before this
    wl = make_workload(name, 0, 0, dtype, dev)
    B, N, model = wl["B"], wl["N"], wl["model"]
    wl["one_pass"]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(timed):
        wl["one_pass"]()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / timed
    res, cin, cout = wl["res"], wl["cin"], wl["cout"]
    h = model._ensure_engine(B, res, dev)
    prof = _lib.UNetProfile()
    xs = torch.randn(B, cin, res, res, device=dev)
    ts = torch.full((B,), 0.5, device=dev)
    od = torch.empty(B, cout, res, res, device=dev)
    _lib.check(lib.bndm_unet_profile(h, C.c_void_p(xs.data_ptr()), C.c_void_p(ts.data_ptr()), C.c_void_p(od.data_ptr()), B, 3,
                                     C.byref(prof), _lib.current_stream_ptr()), "bndm_unet_profile")
    flops = sum(f for _, _, f in engine_ops(h)) * B
    tf = flops / (prof.ms_total * 1e-3) / 1e12
    out = {"workload": wl["workload"], "value": round(B / el, 3), "unit": "images/sec", "batch_per_gpu": B, "nb_steps": N,
           "ms_per_step": round(el * 1e3, 2), "ms_per_forward": round(prof.ms_total, 3), "launches_per_forward": prof.launches,
           "tflops": round(tf, 1), "frac_of_peak": round(tf / PEAK_F16_TFLOPS, 4)}
    if "vae" in wl:
        from bndm_amd.vae import vae_decode
        lat = 0.18215 * torch.randn(B, 4, 64, 64, device=dev)
        vae_decode(wl["vae"], lat)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            vae_decode(wl["vae"], lat)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 2
        out["vae_decode_ms_per_batch"] = round(ms, 2)
        out["vae_decode_tflops"] = round(2.51e12 * B / (ms * 1e-3) / 1e12, 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"], help="BASELINE.json configuration")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per pass (default: the configuration's)")
    ap.add_argument("--nb_steps", type=int, default=0, help="denoising steps per image (default: the configuration's)")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-only", action="store_true", help="skip the timed passes; only the per-kernel profile")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short c3 / c4 / c5 passes that follow the timed c2 region (other_configs)")
    ap.add_argument("--allow-ablation", action="store_true",
                    help="profiling tools only (tools/ablate.sh): run although BNDM_ABLATE is set; the line is marked invalid")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="test hook: exercise the rank launch / barrier / max-over-ranks plumbing on CPU (gloo), no compute")
    argv = sys.argv[1:]
    args = ap.parse_args(argv)

    # switches of the library that are present in the environment are reported with the line; the ablation switch of the
    # profiling build (kernels with work removed) is refused outright
    env_seen = {k: v for k, v in sorted(os.environ.items()) if k.startswith("BNDM_")}
    if "BNDM_ABLATE" in env_seen and not args.allow_ablation:
        raise SystemExit("bench.py: BNDM_ABLATE is set (profiling switch of tools/ablate.sh): refusing to benchmark")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))

    import torch
    from bndm_amd.parallel import barrier, gather_images, init_from_env, max_over_ranks

    rank, world, local = init_from_env("gloo" if args.dry_run_cpu else None)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a mislabelled line")
    if args.dry_run_cpu:
        barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        barrier()
        el = max_over_ranks(time.perf_counter() - t0)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "elapsed": el}), flush=True)
        barrier()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    from bndm_amd import _lib
    from bndm_amd.bluenoise import get_noise_v2
    lib = _lib.load()

    torch.manual_seed(1234 + rank)
    wl = make_workload(args.config, args.batch, args.nb_steps, args.dtype, dev, rank, world)
    B, N, model = wl["B"], wl["N"], wl["model"]
    metric_name, workload_name = wl["metric"], wl["workload"]

    def one_pass():
        return gather_images(wl["one_pass"](), dst=0)

    elapsed = float("nan")
    if not args.profile_only:
        for _ in range(args.warmup):
            one_pass()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_pass()
        barrier()
        torch.cuda.synchronize()
        elapsed = max_over_ranks(time.perf_counter() - t0, device=dev)

    # ---- dominant kernel of the UNet forward, timed with HIP events on the launch stream ---------------------
    roof = None
    if rank == 0:
        prof = _lib.UNetProfile()
        res, cin, cout = wl["res"], wl["cin"], wl["cout"]
        h = model._ensure_engine(B, res, dev)
        xs = torch.randn(B, cin, res, res, device=dev)
        ts = torch.full((B,), 0.5, device=dev)
        od = torch.empty(B, cout, res, res, device=dev)
        rc = lib.bndm_unet_profile(h, C.c_void_p(xs.data_ptr()), C.c_void_p(ts.data_ptr()), C.c_void_p(od.data_ptr()),
                                   B, 3, C.byref(prof), _lib.current_stream_ptr())
        _lib.check(rc, "bndm_unet_profile")
        from bndm_amd.unet import engine_ops
        names = {k for k, _, _ in engine_ops(h)}
        # the engine marks the 256-pixel-tile launches as dominant; batches too small for them run 128-pixel tiles only
        kernel_tag = "conv_t32<TH=16>" if "conv_t32<TH=16>" in names else "conv_t32<TH=8>"
        tile_txt = "256-pixel" if kernel_tag.endswith("16>") else "128-pixel"
        achieved = prof.dom_flops / (prof.ms_dom * 1e-3) / 1e12 if prof.ms_dom > 0 else 0.0
        conv_all = prof.conv_flops / (prof.ms_conv * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": f"{kernel_tag} (GroupNorm+SiLU fused 3x3 conv, {tile_txt} x 128-channel tiles, two workgroups per CU)",
                "achieved": round(achieved, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_F16_TFLOPS, 4),
                # HBM bytes per launch: rocprofv3 PMC passes of this very build (tools/pmc_traffic.py), else null
                "traffic": pmc_traffic(_lib.LIB_PATH, kernel_tag, args.config),
                "launches_per_forward": prof.dom_launches,
                "avg_launch_us": round(prof.ms_dom / max(prof.dom_launches, 1) * 1e3, 2),
                "algorithmic_flop_per_launch": prof.dom_flops / max(prof.dom_launches, 1),
                "algorithmic_bytes_per_launch": prof.dom_bytes / max(prof.dom_launches, 1),
                "all_conv_kernels_tflops": round(conv_all, 1), "ms_per_forward_conv": round(prof.ms_conv, 3),
                "ms_per_forward_total": round(prof.ms_total, 3), "launches_total": prof.launches}
        # the whole timed loop against the same peak: algorithmic flops of every launch of a forward x denoising steps x images / s
        # (the per-launch figure above is that of the dominant kernel alone; this one is what the loop sustains)
        if elapsed == elapsed and elapsed > 0:
            fwd_flops = sum(f for _, _, f in engine_ops(h))                      # per sample
            e2e = fwd_flops * N * (B * args.steps / elapsed) / 1e12
            roof["end_to_end_tflops"] = round(e2e, 1)
            roof["end_to_end_frac"] = round(e2e / PEAK_F16_TFLOPS, 4)

    # ---- per-stage figures: blue-noise transform vs its HBM / fp32-MFMA rooflines -------------
    stages = None
    if rank == 0 and wl["L"] is not None:
        L = wl["L"]

        def time_noise(nb, reps=200):
            """us per bndm_bluenoise call (both kernels), called through the C ABI with caller-owned buffers -- the
            Python wrapper's allocations would dominate a 15-us transform."""
            zz = torch.randn(nb, 3, 64, 64, device=dev)
            outs = [torch.empty_like(zz) for _ in range(3)]
            wsb = lib.bndm_bluenoise_workspace_bytes(nb, 3, 64)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            p = lambda t: C.c_void_p(t.data_ptr())
            stream = _lib.current_stream_ptr()

            def call():
                return lib.bndm_bluenoise(p(L), 0, p(zz), 0, C.c_void_p(0), p(outs[0]), p(outs[1]), p(outs[2]), nb, 0, nb, 3,
                                          64, 1, p(ws), wsb, stream)
            for _ in range(5):
                _lib.check(call(), "bndm_bluenoise")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                call()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        TRI = 4096 * 4097 // 2
        us_small, us_batch = time_noise(2), time_noise(64)
        stages = {
            "bluenoise_B2_us": round(us_small, 1),
            "bluenoise_B2_GBps_of_L": round(4 * TRI / (us_small * 1e-6) / 1e9, 1),      # HBM-bound regime (n = 6 columns)
            "bluenoise_B64_us": round(us_batch, 1),
            "bluenoise_B64_fp32_TFLOPs": round(2.0 * TRI * 192 / (us_batch * 1e-6) / 1e12, 2),   # fp32-MFMA-bound regime
            "hbm_peak_GBps": 8000, "fp32_mfma_peak_TFLOPs": 157.3,
            "unet_ms_per_denoising_step": roof["ms_per_forward_total"] if roof else None,
        }
    if rank == 0 and "vae" in wl:
        from bndm_amd.vae import vae_decode
        lat = 0.18215 * torch.randn(B, 4, 64, 64, device=dev)
        vae_decode(wl["vae"], lat)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            vae_decode(wl["vae"], lat)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        vae_tf = 2.51e12 * B / (ms * 1e-3) / 1e12                  # 2.51 TFLOP per image (tests/test_oracle_vae.py)
        stages = dict(stages or {}, vae_decode_ms_per_batch=round(ms, 2), vae_decode_tflops=round(vae_tf, 1),
                      vae_decode_frac_of_mfma_peak=round(vae_tf / PEAK_F16_TFLOPS, 4))

    # ---- the other BASELINE.json configurations at their per-GPU sizes, short passes after the timed c2 region ------
    others = None
    side_errors = []                 # failures outside the timed region: flagged in the line ("partial": true) and on stderr; the
                                     # exit code stays 0 because the headline value is valid (its timed region ended before)
    if rank == 0 and args.config == "c2" and args.gpus == 1 and not (args.profile_only or args.no_other_configs):
        others = {}
        del wl, model
        torch.cuda.empty_cache()
        for oc in ("c3", "c4", "c5"):
            # a failure in a side configuration must not cost the headline line (the timed region is over)
            try:
                others[oc] = short_pass(oc, args.dtype, dev, lib)
            except Exception as e:                                   # noqa: BLE001 -- reported, not swallowed
                others[oc] = {"error": f"{type(e).__name__}: {e}"[:300]}
                side_errors.append(f"{oc}: {others[oc]['error']}")
                print(f"bench.py: side configuration {oc} FAILED: {others[oc]['error']}", file=sys.stderr, flush=True)
            torch.cuda.empty_cache()
        wl = None

    if rank == 0:
        imgs = args.gpus * B * args.steps
        base = None
        if not (args.no_cpu_baseline or args.gpus > 1 or args.profile_only):
            try:
                base = cpu_baseline()
            except Exception as e:                                   # noqa: BLE001
                base = {"error": f"{type(e).__name__}: {e}"[:300]}
                side_errors.append(f"cpu_baseline: {base['error']}")
                print(f"bench.py: cpu_baseline FAILED: {base['error']}", file=sys.stderr, flush=True)
        line = {
            "metric": metric_name,
            "value": round(imgs / elapsed, 3), "unit": "images/sec", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload_name, "baseline_config": args.config,
                       "global_batch": B * args.gpus, "nb_steps": N, "parallelism": f"batch-shard x{args.gpus}"},
            "roofline": roof,
            "stages": stages,
            # c3 / c4 / c5 at their per-GPU batch (1 warm-up + 2 timed passes each), same library, same process
            "other_configs": others,
            "env": env_seen,
            **({"valid": False, "ablation": env_seen["BNDM_ABLATE"]} if "BNDM_ABLATE" in env_seen else {}),
            # timed on rank 0 of the single-GPU run only (the host cores are shared by the ranks otherwise)
            "cpu_baseline": base,
            **({"partial": True, "side_errors": side_errors} if side_errors else {}),
        }
        print(json.dumps(line), flush=True)
    barrier()


if __name__ == "__main__":
    main()
