"""GPU aid: repeated VAE decodes must be bit-identical (race screen)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bndm_amd.vae import AutoencoderKL
m = AutoencoderKL().cuda()
z = torch.randn(2, 4, 64, 64, device="cuda")
ref = m.decode(z).sample.clone()
bad = 0
for i in range(10):
    y = m.decode(z).sample
    bad += int(not torch.equal(y, ref))
print(f"VAE: {bad} of 10 repeats differ", "max|d|=%g" % (y - ref).abs().max().item())
