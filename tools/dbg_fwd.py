import sys, torch
sys.path.insert(0, "/root/repo")
from bndm_amd import _lib
if len(sys.argv) > 2:
    _lib.LIB_PATH = sys.argv[2]
from bndm_amd.sampler import get_model
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
m = get_model(3, 6, 64, dtype="f16", seed=0).to(dev).eval()
x = torch.randn(B, 3, 64, 64, device=dev)
t = torch.linspace(0.1, 0.9, B, device=dev)
with torch.no_grad():
    y = m(x, t, return_dict=False)[0]
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
