"""Stage figure for SURVEY 8 f1: time of one AutoencoderKL decode (sd-vae-ft-mse layout, synthetic weights) of
B latents 4x64x64 -> 3x512x512 on the HIP path, as TFLOP/s of the decoder's algorithmic 2.51 TFLOP per image."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bndm_amd.vae import AutoencoderKL, vae_decode

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = AutoencoderKL().cuda()
x = 0.18215 * torch.randn(B, 4, 64, 64, device="cuda")
y = vae_decode(m, x)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for _ in range(n):
    y = vae_decode(m, x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"decode B={B}: {dt * 1e3:.1f} ms  ({dt / B * 1e3:.1f} ms/image, {2.5145 * B / dt:.0f} TFLOP/s), out {tuple(y.shape)}, "
      f"finite {bool(torch.isfinite(y).all())}")
