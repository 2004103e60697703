"""Case tables + seeded inputs shared by tests/golden/make_golden.py consumers.

Kept separate from make_golden.py so that tests never import anything that touches
/root/reference (absent on the GPU box)."""
import numpy as np
import torch

STRIDE = 13

NOISE_CASES = []
for _res in (32, 64, 128):
    for _nt in ("gaussianBN", "GBN", "gaussian"):
        if _nt == "gaussian" and _res == 32:
            continue
        for _inplace in (True, False):
            for _tt in ("train", "test"):
                NOISE_CASES.append((_res, _nt, _inplace, _tt))


def case_inputs(seed, B, C, res):
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((B, C, res, res)).astype(np.float32)
    alpha = rs.uniform(0.0, 1.0, size=(B,)).astype(np.float32)
    return x, alpha


def noise_case_shape(res):
    return (3, 3) if res != 128 else (3, 2)


def reference_draw(ci, res, nt, B, C):
    """The white sample the reference drew in its non-inplace branch (CPU generator, seed 77+ci)."""
    torch.manual_seed(77 + ci)
    if nt == "gaussian" or res == 64:
        return torch.randn(B, C, res, res).numpy()
    if res == 32:
        return torch.randn(B, C, 64, 64).numpy()
    return torch.randn(4 * B, C, 64, 64).numpy()          # 128 px: randn(4B, C, 64, 64)


LOOP_CASES = [
    ("gaussian", 3, "linear", (1.0, 0.0, 3.0), 10),
    ("gaussian", 3, "linear", (1.0, 0.0, 3.0), 250),
    ("gaussianBN", 6, "sigmoid", (1000.0, 0.0, 3.0), 10),
    ("gaussianBN", 6, "sigmoid", (1000.0, 0.0, 3.0), 250),
    ("gaussianBN", 6, "sigmoid", (0.2, 0.0, 3.0), 250),
    ("gaussianBN", 3, "linear", (1.0, 0.0, 3.0), 10),
    ("GBN", 3, "linear", (1.0, 0.0, 3.0), 10),
    ("gaussianBN", 6, "sigmoid", (1000.0, 0.0, 3.0), 1000),
]


class FakeModel:
    """Analytic stand-in for the UNet (same formula as in make_golden.py)."""

    def __init__(self, out_channel, C=3):
        self.oc, self.C = out_channel, C

    def __call__(self, x, t, return_dict=False):
        s = t.view(-1, 1, 1, 1)
        d = torch.tanh(x[:, :self.C]) * (0.5 + s) - 0.25 * x[:, :self.C]
        if self.oc == 2 * self.C:
            d = torch.cat([d, torch.cos(3.0 * x[:, :self.C]) * (1.0 - s)], dim=1)
        return (d,)
