"""GPU debugging aid: rel-L2 of the HIP UNet vs the oracle for small configs, fused path on/off."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import unet_oracle as U
from bndm_amd.unet import UNet2DModel

def run(res, boc, cin, cout, B=2, attn_down=None, attn_up=None, seed=3):
    n = len(boc)
    cfg = dict(in_channels=cin, out_channels=cout, block_out_channels=tuple(boc),
               down_attn=tuple(i == attn_down for i in range(n)), up_attn=tuple(i == attn_up for i in range(n)),
               layers_per_block=2)
    sd = U.init_params(cfg, seed=seed, perturb_norm=0.1)
    x = torch.randn(B, cin, res, res, generator=torch.Generator().manual_seed(0))
    t = torch.full((B,), 0.37)
    ref = U.forward(sd, cfg, x, t)
    out = {}
    for nf in ("1", "0"):
        os.environ["BNDM_NO_FUSED"] = nf
        m = UNet2DModel(in_channels=cin, out_channels=cout, block_out_channels=boc,
                        down_block_types=tuple("AttnDownBlock2D" if a else "DownBlock2D" for a in cfg["down_attn"]),
                        up_block_types=tuple("AttnUpBlock2D" if a else "UpBlock2D" for a in cfg["up_attn"]))
        m.load_state_dict(sd)
        m = m.cuda()
        got = m(x.cuda(), t.cuda(), return_dict=False)[0].cpu()
        out[nf] = float((got - ref).double().norm() / ref.double().norm())
        m.release_engine()
    print(f"res={res} boc={boc} B={B}: unfused {out['1']:.2e}  fused {out['0']:.2e}", flush=True)

run(16, (128, 256), 3, 3)            # 16x16 fused (TH=8), 8x8 unfused
run(16, (128, 128), 3, 3)
run(32, (128, 128), 3, 3)            # 32x32 (TH=16) + 16x16 (TH=8)
run(32, (128, 256), 3, 6)
run(64, (128, 128), 3, 3, B=1)
run(64, (128, 128, 256), 3, 3, B=3)
