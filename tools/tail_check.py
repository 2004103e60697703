"""Development check of the <= 8x8 conv_s path: forward vs the fp32 oracle and vs the igemm + gn_small path
(BNDM_NO_TAIL=1), plus the per-op event profile at B=64.  Usage: python tools/tail_check.py [small|big|prof]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet_oracle as U                      # noqa: E402
from bndm_amd.unet import UNet2DModel                    # noqa: E402


def build(res, cin, cout, dtype="f16", no_tail=False, latent=False):
    cfg = U.make_config(res, cin, cout, latent=latent)
    sd = U.init_params(cfg, seed=3, perturb_norm=0.1)
    m = UNet2DModel(in_channels=cin, out_channels=cout, block_out_channels=cfg["block_out_channels"],
                    down_block_types=tuple("AttnDownBlock2D" if a else "DownBlock2D" for a in cfg["down_attn"]),
                    up_block_types=tuple("AttnUpBlock2D" if a else "UpBlock2D" for a in cfg["up_attn"]), dtype=dtype)
    m.load_state_dict(sd)
    m = m.to("cuda").eval()
    m._no_tail = no_tail
    return m, cfg, sd


def fwd(m, x, t):
    # the engine is finalised at the first forward: the switch must be in the environment then
    if m._no_tail:
        os.environ["BNDM_NO_TAIL"] = "1"
    else:
        os.environ.pop("BNDM_NO_TAIL", None)
    return m(x, t, return_dict=False)[0]


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


mode = sys.argv[1] if len(sys.argv) > 1 else "small"
if mode == "small":
    for res, cin, cout, B in [(64, 3, 6, 2), (64, 3, 3, 5), (128, 3, 6, 1)]:
        x = torch.randn(B, cin, res, res, generator=torch.Generator().manual_seed(0))
        t = torch.linspace(0.9, 0.2, B)
        m, cfg, sd = build(res, cin, cout)
        t0 = time.time()
        ref = U.forward(sd, cfg, x, t)
        got = fwd(m, x.cuda(), t.cuda()).cpu()
        m0, _, _ = build(res, cin, cout, no_tail=True)
        old = fwd(m0, x.cuda(), t.cuda()).cpu()
        print(f"res{res} {cin}->{cout} B={B}: new vs oracle {rel(got, ref):.3e}   old vs oracle {rel(old, ref):.3e}   "
              f"new vs old {rel(got, old):.3e}  ({time.time() - t0:.1f}s)", flush=True)
        got2 = fwd(m, x.cuda(), t.cuda()).cpu()
        print("   repeatable:", torch.equal(got, got2), "ops:", len(m.engine_ops(B, res, torch.device("cuda", 0))))
elif mode == "big":
    x = torch.randn(64, 3, 64, 64, generator=torch.Generator().manual_seed(0)).cuda()
    t = torch.linspace(0.95, 0.05, 64).cuda()
    m, cfg, sd = build(64, 3, 6)
    m0, _, _ = build(64, 3, 6, no_tail=True)
    a = fwd(m, x, t)
    b = fwd(m0, x, t)
    print("B=64 new vs old", rel(a.cpu(), b.cpu()), flush=True)
    ref = U.forward(sd, cfg, x[:2].cpu(), t[:2].cpu())
    print("B=64 rows 0..1 vs oracle: new", rel(a[:2].cpu(), ref), "old", rel(b[:2].cpu(), ref))
    for mm, nm in ((m, "new"), (m0, "old")):
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(20):
            mm(x, t, return_dict=False)
        torch.cuda.synchronize()
        print(nm, "ms/forward", (time.time() - t0) / 20 * 1e3)
