"""CPU tests of the on-disk formats of the path and of the loaders' key handling."""
import os

import numpy as np
import pytest
import torch


def test_factor_npz_roundtrip_key_x(tmp_path):
    """np.load(path)['x'].astype(float32) (iadb_bn.py:83-86): the factor file is a .npz with key 'x', any float dtype."""
    from bndm_amd.synth import formula_factor, load_or_make_factor
    L = formula_factor()
    p64 = str(tmp_path / "cov_gaussianBN_L_res64_d3.npz")
    np.savez(p64, x=L.astype(np.float64))                                # the file dtype is cast on load
    got = load_or_make_factor(p64, "blue")
    assert got.dtype == np.float32 and got.shape == (4096, 4096)
    assert np.array_equal(got, L)
    assert np.array_equal(np.triu(got, 1), np.zeros_like(got))          # lower-triangular container survives
    pc = str(tmp_path / "compressed.npz")
    np.savez_compressed(pc, x=L)
    assert np.array_equal(load_or_make_factor(pc), L)
    with pytest.raises(KeyError):
        np.savez(str(tmp_path / "bad.npz"), y=L[:4])
        load_or_make_factor(str(tmp_path / "bad.npz"))


def test_missing_factor_falls_back_to_the_documented_synthetic_one(tmp_path):
    from bndm_amd.synth import load_or_make_factor
    L = load_or_make_factor(str(tmp_path / "absent.npz"), "blue")
    assert L.shape == (4096, 4096) and L.dtype == np.float32
    assert np.allclose((L.astype(np.float64) ** 2).sum(1)[::257], 1.0, atol=1e-4)     # unit-variance rows


def test_scheduler_params_txt(tmp_path):
    """np.loadtxt(f'{output_folder}/scheduler_params.txt') (iadb_bn.py:735) -> (tau, start, end)."""
    p = tmp_path / "scheduler_params.txt"
    np.savetxt(str(p), np.array([1000.0, 0.0, 3.0]))
    sp = np.loadtxt(str(p))
    from bndm_amd.schedules import step_tables
    t_in, da, dg = step_tables(250, "linear", "sigmoid", torch.from_numpy(sp).float())
    assert t_in.shape == da.shape == dg.shape == (250,)
    assert abs(float(dg.sum()) - 1.0) < 1e-5 and abs(float(da.sum()) - 1.0) < 1e-5


def test_deprecated_attention_keys_are_renamed():
    """Checkpoints from diffusers < 0.18 (sd-vae-ft-mse; latent_iadb_bn_diffusers.py:70) name attention weights
    query / key / value / proj_attn, some as 1x1 convs: both loaders accept them."""
    from bndm_amd.unet import UNet2DModel
    from bndm_amd.vae import AutoencoderKL
    m = UNet2DModel(in_channels=3, out_channels=3, block_out_channels=(64, 64),
                    down_block_types=("DownBlock2D", "AttnDownBlock2D"), up_block_types=("AttnUpBlock2D", "UpBlock2D"),
                    seed=1)
    new = {k: v.clone() for k, v in m.state_dict().items()}
    ren = ((".to_q.", ".query."), (".to_k.", ".key."), (".to_v.", ".value."), (".to_out.0.", ".proj_attn."))
    old = {}
    for k, v in new.items():
        ok = k
        for a, b in ren:
            ok = ok.replace(a, b)
        if ok != k and k.endswith(".weight"):
            v = v[:, :, None, None]                                      # conv-shaped projection weights
        old[ok] = v
    assert any(".query." in k for k in old) and not any(".to_q." in k for k in old)
    m2 = UNet2DModel(in_channels=3, out_channels=3, block_out_channels=(64, 64),
                     down_block_types=("DownBlock2D", "AttnDownBlock2D"), up_block_types=("AttnUpBlock2D", "UpBlock2D"),
                     seed=2)
    m2.load_state_dict(old)                                              # strict
    for k, v in m2.state_dict().items():
        assert torch.equal(v, new[k]), k
    v1 = AutoencoderKL(block_out_channels=(128, 256), layers_per_block=1, seed=3)
    newv = {k: t.clone() for k, t in v1.state_dict().items()}
    oldv = {}
    for k, t in newv.items():
        ok = k
        for a, b in ren:
            ok = ok.replace(a, b)
        oldv[ok] = t
    oldv["encoder.conv_in.weight"] = torch.zeros(1)                      # full-autoencoder checkpoints carry these
    v2 = AutoencoderKL(block_out_channels=(128, 256), layers_per_block=1, seed=4)
    v2.load_state_dict(oldv)
    for k, t in v2.state_dict().items():
        assert torch.equal(t, newv[k]), k


def test_unet_create_rejects_unsupported_first_width():
    from bndm_amd import _lib
    import ctypes as C
    lib = _lib.load()
    cfg = _lib.UNetConfig()
    cfg.in_channels, cfg.out_channels, cfg.resolution, cfg.num_levels = 3, 3, 64, 2
    cfg.block_out_channels[0], cfg.block_out_channels[1] = 192, 192
    cfg.layers_per_block, cfg.dtype, cfg.max_batch = 2, 0, 1
    h = C.c_void_p()
    rc = lib.bndm_unet_create(C.byref(h), C.byref(cfg))
    assert rc == -1 and b"block_out_channels[0]" in lib.bndm_last_error()
