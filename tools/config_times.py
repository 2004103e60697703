"""Supplementary timings of the other BASELINE.json configurations on ONE MI355X (synthetic weights / data):
   c3  church_res64 DDIM, 100 steps, B=64, UNet 3->3
   c4  celeba_res128 IADB, 250 steps, out 6, sigmoid(0.2,0,3): the per-GPU share B=32 of the 8-GPU batch of 256
   c5  latent cat_res512 IADB, 250 steps, UNet 4->8: the per-GPU share B=8, + VAE decode to 512x512
Prints images/s per configuration (wall clock around sample + export, after one warm-up call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bndm_amd.sampler import get_model, sample_iadb, export_u8
from bndm_amd.schedulers import DDIMScheduler, IADBScheduler
from bndm_amd.bluenoise import get_noise_v2
from bndm_amd.synth import load_or_make_factor
from bndm_amd.unet import UNet2DModel
from bndm_amd.vae import AutoencoderKL, vae_decode

dev = torch.device("cuda")
L = torch.from_numpy(load_or_make_factor("__none__.npz", "blue")).to(dev)


def timed(fn, n=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def c3():
    m = get_model(3, 3, 64).to(dev).eval()
    sch = DDIMScheduler()
    sch.set_timesteps(100)
    B = 64
    def run():
        x = torch.randn(B, 3, 64, 64, device=dev)
        return export_u8(sch.sample(m, x), "round")
    dt = timed(run)
    print(f"c3 DDIM res64 100 steps B={B}: {B / dt:.1f} images/s ({dt * 1e3:.0f} ms)")
    m.release_engine()


def c4():
    m = get_model(3, 6, 128).to(dev).eval()
    B = 32
    params = torch.tensor([0.2, 0.0, 3.0], device=dev)
    def run():
        x = torch.randn(B, 3, 128, 128, device=dev)
        x0, _, _ = get_noise_v2(dev, x, L, torch.ones(B, device=dev), None, "gaussianBN", "test", True)
        s, _, _ = sample_iadb(m, x0, 250, "sigmoid", params, 6, "gaussianBN", "test")
        return export_u8(s, "trunc")
    dt = timed(run, 1)
    print(f"c4 IADB res128 250 steps B={B} (per-GPU share of 256): {B / dt:.2f} images/s ({dt * 1e3:.0f} ms)")
    m.release_engine()


def c5():
    boc = (128, 128, 256, 256, 512, 512)
    m = UNet2DModel(sample_size=64, in_channels=4, out_channels=8, layers_per_block=2, block_out_channels=boc,
                    down_block_types=tuple("AttnDownBlock2D" if i == 4 else "DownBlock2D" for i in range(6)),
                    up_block_types=tuple("AttnUpBlock2D" if i == 1 else "UpBlock2D" for i in range(6))).to(dev).eval()
    vae = AutoencoderKL().to(dev).eval()
    sch = IADBScheduler(noise_type="gaussianBN", out_channels=8)
    sch.set_timesteps(250)
    B = 8
    def loop():
        return sch.sample(m, torch.randn(B, 4, 64, 64, device=dev))
    def full():
        return export_u8(vae_decode(vae, loop()), "round")
    dl, df = timed(loop, 1), timed(full, 1)
    print(f"c5 latent IADB 250 steps B={B} (per-GPU share of 64): loop {B / dl:.2f} latents/s ({dl * 1e3:.0f} ms); "
          f"with VAE decode to 512x512: {B / df:.2f} images/s ({df * 1e3:.0f} ms)")


if __name__ == "__main__":
    for f in ((c3, c4, c5) if len(sys.argv) < 2 else [globals()[a] for a in sys.argv[1:]]):
        f()
