"""GPU aid: the same forward repeated N times must be bit-identical (race screen for in-launch hand-offs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bndm_amd.sampler import get_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = get_model(3, 6, 64, seed=0).cuda()
x = torch.randn(B, 3, 64, 64, device="cuda")
t = torch.full((B,), 0.5, device="cuda")
ref = m(x, t, return_dict=False)[0].clone()
bad = 0
for i in range(40):
    y = m(x, t, return_dict=False)[0]
    if not torch.equal(y, ref):
        bad += 1
print(f"B={B}: {bad} of 40 repeats differ", "max|d|=%g" % (y - ref).abs().max().item())
