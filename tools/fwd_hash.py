"""Profiling aid: sha256 of one UNet forward's output under a given build of the library, on fixed inputs -- two builds that
are meant to be bit-identical (a rescheduled kernel) must print the same line:
    python tools/fwd_hash.py tools/libbndm_prev.so [c2|c3|c4|c5]"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bndm_amd import _lib   # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch                # noqa: E402
import bench                # noqa: E402

name = sys.argv[2] if len(sys.argv) > 2 else "c2"
dev = torch.device("cuda:0")
wl = bench.make_workload(name, 0, 0, "f16", dev)
B, res, cin = wl["B"], wl["res"], wl["cin"]
g = torch.Generator(device="cpu").manual_seed(7)
x = torch.randn(B, cin, res, res, generator=g).to(dev)
t = torch.linspace(0.05, 0.95, B).to(dev)
with torch.no_grad():
    y = wl["model"](x, t)
    y = y.sample if hasattr(y, "sample") else y
torch.cuda.synchronize()
print(name, hashlib.sha256(y.float().cpu().numpy().tobytes()).hexdigest()[:16], float(y.float().abs().mean()))
