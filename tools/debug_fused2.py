import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import unet_oracle as U
from bndm_amd.sampler import get_model
cfg = U.make_config(64, 3, 6)
sd = U.init_params(cfg, seed=3, perturb_norm=0.1)
x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(0))
t = torch.tensor([0.996, 0.4])
ref = U.forward(sd, cfg, x, t)
for env in ({"BNDM_NO_FUSED": "1"}, {"BNDM_FUSED_MIN": "64"}, {"BNDM_FUSED_MIN": "32", "BNDM_FUSED_MAX": "32"},
            {"BNDM_FUSED_MIN": "16", "BNDM_FUSED_MAX": "16"}, {}):
    for k in ("BNDM_NO_FUSED", "BNDM_FUSED_MIN", "BNDM_FUSED_MAX"):
        os.environ.pop(k, None)
    os.environ.update(env)
    m = get_model(3, 6, 64)
    m.load_state_dict(sd)
    m = m.cuda()
    got = m(x.cuda(), t.cuda(), return_dict=False)[0].cpu()
    per = [(float((got[i] - ref[i]).double().norm() / ref[i].double().norm())) for i in range(2)]
    print(env, "rel-L2 per sample", ["%.2e" % p for p in per], flush=True)
    m.release_engine()
