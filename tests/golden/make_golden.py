#!/usr/bin/env python
"""Generate golden vectors by importing the REFERENCE (read-only, /root/reference) on CPU.

Run in the build container only (the reference does not exist on the GPU box):

    MPLBACKEND=Agg python tests/golden/make_golden.py

Outputs (committed, data only -- inputs are regenerated from seeds on the consumer side):
  noise_cases.npz   get_noise_v2 outputs for {32,64,128} px x {gaussian,gaussianBN,GBN} x
                    {inplace, non-inplace} x {train,test}; flat[::STRIDE] subsamples + float64 sums
  schedules.npz     gamma / alpha tables from utils.get_scheduler_gamma / get_scheduler
  loops.npz         utils.sample_iadb trajectories with an analytic fake model
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from bndm_amd.synth import formula_factor  # noqa: E402

STRIDE = 13


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_noise_module():
    return _load("ref_get_noise_recent", os.path.join(REF, "bluenoise", "get_noise_recent.py"))


def ref_utils_module():
    stub = types.ModuleType("diffusers")
    stub.UNet2DModel = object
    sys.modules.setdefault("diffusers", stub)
    return _load("ref_utils", os.path.join(REF, "utils.py"))


def case_inputs(seed, B, C, res):
    """Shared by generator and consumers: legacy MT19937 stream, frozen by numpy."""
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((B, C, res, res)).astype(np.float32)
    alpha = rs.uniform(0.0, 1.0, size=(B,)).astype(np.float32)
    return x, alpha


NOISE_CASES = []
for res in (32, 64, 128):
    for nt in ("gaussianBN", "GBN", "gaussian"):
        if nt == "gaussian" and res == 32:
            continue                        # NotImplementedError in the reference (:58-59)
        for inplace in (True, False):
            for tt in ("train", "test"):
                NOISE_CASES.append((res, nt, inplace, tt))


def noise_goldens():
    ref = ref_noise_module()
    out = {}
    for lname, L in (("formula", formula_factor()), ("identity", np.eye(4096, dtype=np.float32))):
        Lt = torch.from_numpy(L)
        for ci, (res, nt, inplace, tt) in enumerate(NOISE_CASES):
            if lname == "identity" and not (res == 128 and inplace):
                continue                    # identity L only to expose the 128-px permutation
            B, C = (3, 3) if res != 128 else (3, 2)
            x, alpha = case_inputs(1000 + ci, B, C, res)
            torch.manual_seed(77 + ci)
            n, nb, nw = ref.get_noise_v2(torch.device("cpu"), torch.from_numpy(x.copy()), Lt,
                                         torch.from_numpy(alpha), None, noise_type=nt,
                                         train_or_test=tt, inplace=inplace)
            key = f"{lname}|{res}|{nt}|{int(inplace)}|{tt}"
            for tag, arr in (("n", n), ("bn", nb), ("wn", nw)):
                a = arr.contiguous().numpy().astype(np.float32)
                out[f"{key}|{tag}"] = a.reshape(-1)[::STRIDE].copy()
                out[f"{key}|{tag}|sum"] = np.array([a.astype(np.float64).sum(),
                                                    np.abs(a.astype(np.float64)).sum()])
                out[f"{key}|{tag}|shape"] = np.array(a.shape)
    np.savez_compressed(os.path.join(HERE, "noise_cases.npz"), **out)
    print("noise_cases:", len(out), "arrays")


def schedule_goldens():
    ru = ref_utils_module()
    out = {}
    for N in (250, 1000, 100, 50):
        t = torch.arange(0, N + 1).float()
        out[f"alpha|linear|{N}"] = ru.get_scheduler(t, "linear", N).numpy()
        out[f"gamma|linear|{N}"] = ru.get_scheduler_gamma(t, "linear", [1.0, 0.0, 3.0], N).numpy()
        for params in ((1000.0, 0.0, 3.0), (0.2, 0.0, 3.0), (0.9, 0.5, 3.0)):
            p = torch.tensor(params)
            out[f"gamma|sigmoid|{N}|{params}"] = ru.get_scheduler_gamma(t, "sigmoid", p, N).numpy()
        p = torch.tensor((2.0, 0.1, 0.9))
        out[f"gamma|cosine|{N}|(2.0, 0.1, 0.9)"] = ru.get_scheduler_gamma(t, "cosine", p, N).numpy()
    np.savez_compressed(os.path.join(HERE, "schedules.npz"), **out)
    print("schedules:", len(out), "arrays")


class FakeModel:
    """Analytic stand-in for the UNet: d = tanh(x)*s(t) stacked to out_channel maps."""

    def __init__(self, out_channel, C=3):
        self.oc, self.C = out_channel, C

    def __call__(self, x, t, return_dict=False):
        s = t.view(-1, 1, 1, 1)
        d = torch.tanh(x[:, :self.C]) * (0.5 + s) - 0.25 * x[:, :self.C]
        if self.oc == 2 * self.C:
            d = torch.cat([d, torch.cos(3.0 * x[:, :self.C]) * (1.0 - s)], dim=1)
        return (d,)


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


def loop_goldens():
    ru = ref_utils_module()
    out = {}
    for ci, (nt, oc, gs, params, N) in enumerate(LOOP_CASES):
        x0, _ = case_inputs(2000 + ci, 2, 3, 8)
        model = FakeModel(oc)
        xa, xall, _ = ru.sample_iadb(model, torch.from_numpy(x0.copy()), N, gs, torch.tensor(params),
                                     oc, nt, "test")
        key = f"{nt}|{oc}|{gs}|{params}|{N}"
        out[key + "|final"] = xa.numpy()
        out[key + "|nsnap"] = np.array(len(xall))
        out[key + "|snap_first"] = xall[0].numpy()
        out[key + "|snap_mid"] = xall[len(xall) // 2].numpy()
        xt = ru.sample_iadb(model, torch.from_numpy(x0.copy()), N, gs, torch.tensor(params), oc, nt, "train")
        assert torch.equal(xt, xa)
    np.savez_compressed(os.path.join(HERE, "loops.npz"), **out)
    print("loops:", len(out), "arrays")


if __name__ == "__main__":
    os.environ.setdefault("MPLBACKEND", "Agg")
    noise_goldens()
    schedule_goldens()
    loop_goldens()
