"""Consumer of tests/golden/diffusers_cases.npz -- the fixture tests/golden/make_diffusers_golden.py writes from the REAL
``diffusers`` classes (UNet2DModel, DDIMScheduler, AutoencoderKL) wherever that package is importable.

In the build image it is not (no wheel, no network), the fixture does not exist, and these tests SKIP: the network / DDIM /
VAE oracles stay "parity unpinned" (DESIGN.md section 2).  The day the fixture is committed they run without any other
change and pin oracle/unet_oracle.py, oracle/sampler_oracle.py::ddim_* and oracle/vae_oracle.py against the dependency
the reference actually calls (iadb_bn.py:282,319; ddim_diffusers.py:499-503,680; latent_iadb_bn_diffusers.py:70,188)."""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GEN = os.path.join(HERE, "golden", "make_diffusers_golden.py")
FIX = os.path.join(HERE, "golden", "diffusers_cases.npz")


def _gen():
    spec = importlib.util.spec_from_file_location("make_diffusers_golden", GEN)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _fixture():
    if not os.path.exists(FIX):
        pytest.skip("tests/golden/diffusers_cases.npz absent (diffusers not installable here): parity unpinned")
    return np.load(FIX)


def _check(fx, key, y, tol):
    g = _gen()
    y = y.detach().to(torch.float64).contiguous().view(-1)
    assert int(fx[key + "/shape"]) == y.numel(), key
    ref = torch.from_numpy(fx[key + "/sub"]).double()
    got = y[::g.STRIDE]
    rel = float((got - ref).norm() / ref.norm())
    s = fx[key + "/sum"]
    assert rel <= tol, (key, rel)
    assert abs(float(y.sum()) - s[0]) <= tol * max(1.0, float(y.abs().sum()))
    assert abs(float((y * y).sum()) - s[1]) <= 10 * tol * s[1]


def test_generator_script_is_inert_without_diffusers():
    """The committed generator runs here, says why it writes nothing and leaves no file behind; its seeded inputs are
    deterministic (the consumer regenerates them)."""
    try:
        import diffusers  # noqa: F401
        pytest.skip("diffusers is importable: run the generator and commit the fixture instead")
    except ImportError:
        pass
    had = os.path.exists(FIX)
    r = subprocess.run([sys.executable, GEN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "not importable" in r.stdout
    assert os.path.exists(FIX) == had
    g = _gen()
    for name, res, cin, cout, latent, hw, batch, tvals in g.UNET_CASES:
        x1, t1 = g.unet_inputs(name, cin, hw, batch, tvals)
        x2, t2 = g.unet_inputs(name, cin, hw, batch, tvals)
        assert torch.equal(x1, x2) and torch.equal(t1, t2) and x1.shape == (batch, cin, hw, hw)


def test_unet_oracle_matches_diffusers_fixture():
    fx = _fixture()
    from oracle import unet_oracle as U
    g = _gen()
    torch.set_num_threads(8)
    for name, res, cin, cout, latent, hw, batch, tvals in g.UNET_CASES:
        cfg = U.make_config(res, cin, cout, latent=latent)
        sd = U.init_params(cfg, seed=21, perturb_norm=0.1)
        x, t = g.unet_inputs(name, cin, hw, batch, tvals)
        y = U.forward(sd, cfg, x, t if t.numel() > 1 else t[0])
        _check(fx, "unet/" + name, y, 2e-5)               # fp32 both sides, different op order only


def test_ddim_oracle_matches_diffusers_fixture():
    fx = _fixture()
    from oracle import sampler_oracle as S
    g = _gen()
    x, eps = g.ddim_inputs()
    for n in g.DDIM_STEPS:
        acp, timesteps, ratio = S.ddim_tables(num_inference=n)
        assert [int(v) for v in timesteps] == [int(v) for v in fx[f"ddim/{n}/timesteps"]]
        ts = [int(v) for v in timesteps]
        for t in (ts[0], ts[len(ts) // 2], ts[-1]):
            _check(fx, f"ddim/{n}/step{t}", S.ddim_step(eps, t, x, acp, ratio), 1e-6)


def test_vae_oracle_matches_diffusers_fixture():
    fx = _fixture()
    from oracle import vae_oracle as V
    g = _gen()
    cfg = V.make_config(g.VAE_CASE["block_out_channels"], g.VAE_CASE["layers_per_block"])
    sd = V.init_params(cfg, seed=22, perturb_norm=0.1)
    _check(fx, "vae/decode", V.vae_decode(sd, cfg, g.vae_input()), 2e-5)
