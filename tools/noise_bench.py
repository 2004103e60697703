"""Blue-noise transform in its HBM regime: per-kernel durations come from rocprofv3 --kernel-trace (run under it)."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from bndm_amd.bluenoise import get_noise_v2
from bndm_amd.synth import blue_noise_factor
dev = torch.device("cuda")
L = torch.from_numpy(blue_noise_factor("blue")).to(dev)
for nb in (2, 10, 64):
    zz = torch.randn(nb, 3, 64, 64, device=dev)
    aa = torch.ones(nb, device=dev)
    for _ in range(20):
        get_noise_v2(dev, zz, L, aa, None, noise_type="GBN", train_or_test="test", inplace=True, l_is_triangular=True)
torch.cuda.synchronize()
