"""GPU parity of (1) the in-engine DDIM loop ``bndm_unet_sample_ddim`` -- what ``cli_ddim.py`` and ``bench.py --config c3``
run (reference loop: ddim_diffusers.py:674-681), (2) the <= 8x8 section on ``conv_s`` (csrc/unet_tail.hip; layers
down_blocks.3-5 / mid_block / up_blocks.0-2 of the network built at iadb_bn.py:209-228) with the kernel names asserted from
``bndm_unet_op_info``, and (3) the chained c4 / c5 paths at their per-GPU sizes (iadb_bn.py:770-790,
latent_iadb_bn_diffusers.py:502-540).

Tolerances: one forward rel-L2 <= 2e-3 (f16); an n-step loop in f16 <= 3e-3; the fp32-compute mode <= 1e-4;
final images PSNR >= 35 dB."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def _pair(res, cin, cout, dtype="f16", latent=False, seed=3):
    from oracle import unet_oracle as U
    from bndm_amd.unet import UNet2DModel
    cfg = U.make_config(res, cin, cout, latent=latent)
    sd = U.init_params(cfg, seed=seed, perturb_norm=0.1)
    m = UNet2DModel(in_channels=cin, out_channels=cout, block_out_channels=cfg["block_out_channels"],
                    down_block_types=tuple("AttnDownBlock2D" if a else "DownBlock2D" for a in cfg["down_attn"]),
                    up_block_types=tuple("AttnUpBlock2D" if a else "UpBlock2D" for a in cfg["up_attn"]), dtype=dtype)
    m.load_state_dict(sd)
    return m.to("cuda").eval(), U, cfg, sd


def _ops(m, B, res):
    return m.engine_ops(B, res, torch.device("cuda", torch.cuda.current_device()))


# ------------------------------------------------------------------------------------------------ DDIM, in-engine loop
@pytest.mark.parametrize("dtype,tol", [("f16", 3e-3), ("f32", 1e-4)])
def test_ddim_in_engine_loop_matches_oracle_loop(dtype, tol):
    """DDIMScheduler.sample -> bndm_unet_sample_ddim (per-schedule time-embedding table for the integer timesteps
    750, 500, 250, 0) against the oracle's loop with the oracle network."""
    from oracle import sampler_oracle as S
    from bndm_amd.schedulers import DDIMScheduler
    m, U, cfg, sd = _pair(64, 3, 3, dtype)
    x0 = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(6))
    ref = S.sample_ddim(U.OracleUNet(cfg, sd), x0.clone(), num_inference=4)
    sch = DDIMScheduler(num_train_timesteps=1000, beta_schedule="linear")
    sch.set_timesteps(4)
    assert [int(t) for t in sch.timesteps] == [750, 500, 250, 0]
    got = sch.sample(m, x0.cuda())
    r = _rel(got.cpu(), ref)
    print(f"DDIM 4-step in-engine loop {dtype}: rel-L2 {r:.3e}")
    assert r <= tol
    # the in-engine loop is the step-by-step loop of ddim_diffusers.py:674-681, bit for bit
    xs = x0.cuda()
    for t in sch.timesteps:
        xs = sch.step(m(xs, int(t)).sample, int(t), xs).prev_sample
    assert torch.equal(got, xs)


def test_ddim_in_engine_loop_at_the_benched_batch():
    """c3's shape (B=64, 3 -> 3): the loop runs conv_t32<TH=16> and conv_s and agrees with the oracle loop."""
    from oracle import sampler_oracle as S
    from bndm_amd.schedulers import DDIMScheduler
    torch.set_num_threads(32)
    m, U, cfg, sd = _pair(64, 3, 3)
    kinds = {(k, lab.split()[-1]) for k, lab, _ in _ops(m, 64, 64)}
    assert ("conv_t32<TH=16>", "64x64") in kinds and ("conv_s<TM=128>", "8x8") in kinds, sorted(kinds)
    x0 = torch.randn(64, 3, 64, 64, generator=torch.Generator().manual_seed(7))
    ref = S.sample_ddim(U.OracleUNet(cfg, sd), x0.clone(), num_inference=2)
    sch = DDIMScheduler(num_train_timesteps=1000, beta_schedule="linear")
    sch.set_timesteps(2)
    got = sch.sample(m, x0.cuda())
    r = _rel(got.cpu(), ref)
    print(f"DDIM 2-step loop B=64: rel-L2 {r:.3e}")
    assert r <= 3e-3


# ------------------------------------------------------------------------------------------------ conv_s
def test_tail_runs_on_conv_s_with_few_launches():
    """B=64 res64: every convolution at 8x8 / 4x4 / 2x2 is ONE conv_s launch, attention is q|k|v+softmax.v + to_out,
    GroupNorm launches remain only where the groups straddle the concatenated tensors (768 channels)."""
    m, U, cfg, sd = _pair(64, 3, 6)
    ops = _ops(m, 64, 64)
    small = [(k, lab) for k, lab, _ in ops if lab.split()[-1] in ("8x8", "4x4", "2x2")]
    kinds = {(k, lab.split()[-1]) for k, lab in small}
    assert ("conv_s<TM=128>", "8x8") in kinds and ("conv_s<TM=64>", "4x4") in kinds and ("conv_s<TM=64>", "2x2") in kinds
    assert ("conv_s<qkv+attention>", "4x4") in kinds and ("conv_s<qkv+attention>", "2x2") in kinds
    assert not any(k in ("conv_igemm", "splitk_reduce", "attention") for k, _ in small), sorted(kinds)
    gn = [lab for k, lab in small if k == "gn_small"]
    assert len(gn) == 2 and all("C=768" in lab for lab in gn), gn
    assert len(small) <= 55 and len(ops) <= 100, (len(small), len(ops))


@pytest.mark.parametrize("res,cin,cout,B,latent", [(64, 3, 6, 3, False), (64, 4, 8, 8, True), (128, 3, 6, 2, False)])
def test_tail_path_matches_oracle_and_the_igemm_path(res, cin, cout, B, latent, monkeypatch):
    """conv_s against the oracle and against the implicit-GEMM + gn_small formulation of the same layers
    (BNDM_NO_TAIL=1, read when the engine is finalised): ragged batches exercise the masked rows of the last tile."""
    m, U, cfg, sd = _pair(res, cin, cout, latent=latent)
    x = torch.randn(B, cin, res, res, generator=torch.Generator().manual_seed(0))
    t = torch.linspace(0.9, 0.1, B)
    ref = U.forward(sd, cfg, x, t)
    got = m(x.cuda(), t.cuda(), return_dict=False)[0].cpu()
    assert any(k.startswith("conv_s") for k, _, _ in _ops(m, B, res))
    monkeypatch.setenv("BNDM_NO_TAIL", "1")
    m0, _, _, _ = _pair(res, cin, cout, latent=latent)
    old = m0(x.cuda(), t.cuda(), return_dict=False)[0].cpu()
    assert not any(k.startswith("conv_s") for k, _, _ in _ops(m0, B, res))
    r_new, r_old, r_x = _rel(got, ref), _rel(old, ref), _rel(got, old)
    print(f"res{res} {cin}->{cout} B={B}: conv_s {r_new:.3e}  igemm {r_old:.3e}  between {r_x:.3e}")
    assert r_new <= 2e-3 and r_old <= 2e-3 and r_x <= 2e-3


def test_tail_bf16_and_repeatability():
    m, U, cfg, sd = _pair(64, 3, 6, "bf16")
    x = torch.randn(4, 3, 64, 64, generator=torch.Generator().manual_seed(2)).cuda()
    t = torch.full((4,), 0.6, device="cuda")
    ref = U.forward(sd, cfg, x.cpu(), t.cpu())
    a = m(x, t, return_dict=False)[0].clone()
    assert _rel(a.cpu(), ref) <= 1e-2
    for _ in range(3):
        assert torch.equal(m(x, t, return_dict=False)[0], a)


# ------------------------------------------------------------------------------------------------ chained paths at size
def test_c4_chain_res128_B32_noise_then_loop(formula_L):
    """BASELINE config 4 per GPU: x0 = get_noise_v2(..., 'gaussianBN', 'test', inplace=True) on 32 x 3 x 128 x 128,
    then a 4-step IADB loop with sigmoid(0.2, 0, 3) -- against the oracle chain (iadb_bn.py:770-790)."""
    from oracle import noise_oracle as NO
    from oracle import sampler_oracle as S
    from bluenoise.get_noise_recent import get_noise_v2
    from utils import sample_iadb
    torch.set_num_threads(32)
    B, N = 32, 4
    m, U, cfg, sd = _pair(128, 3, 6)
    kinds = {(k, lab.split()[-1]) for k, lab, _ in _ops(m, B, 128)}
    assert ("conv_t32<TH=16>", "128x128") in kinds, sorted(kinds)
    z = torch.randn(B, 3, 128, 128, generator=torch.Generator().manual_seed(11))
    params = torch.tensor([0.2, 0.0, 3.0])
    gamma_T = S.gamma_schedule(torch.full((B,), float(N)), N, "sigmoid", params)
    ref_x0 = NO.get_noise_v2(z.numpy(), formula_L, gamma_T.numpy(), "gaussianBN", "test")[0]
    L = torch.from_numpy(formula_L).cuda()
    x0 = get_noise_v2(torch.device("cuda"), z.cuda(), L, gamma_T.cuda(), None, "gaussianBN", "test", True)[0]
    assert np.abs(x0.cpu().numpy() - ref_x0).max() <= 1e-4 * max(1.0, np.abs(ref_x0).max())
    ref = S.sample_iadb(U.OracleUNet(cfg, sd), torch.from_numpy(np.ascontiguousarray(ref_x0)), N, "sigmoid", params, 6,
                        "gaussianBN", "train")
    got = sample_iadb(m, x0.contiguous(), N, "sigmoid", params.cuda(), 6, "gaussianBN", "train")
    r = _rel(got.cpu(), ref)
    print(f"c4 chain (noise -> 4 steps) B=32 res128: rel-L2 {r:.3e}")
    assert r <= 3e-3


def test_c5_chain_latent_noise_loop_decode_export(formula_L):
    """BASELINE config 5 per GPU: GBN latent noise (4 x 64 x 64) -> 4 IADB steps at B=8 (4 -> 8 channels) -> vae_decode of
    two of the latents -> uint8 images ('round', latent_iadb_bn_diffusers.py:539-540) against the oracle chain."""
    from oracle import noise_oracle as NO
    from oracle import sampler_oracle as S
    from oracle import vae_oracle as V
    from bluenoise.get_noise_recent import get_noise_v2
    from bndm_amd.schedulers import IADBScheduler
    from bndm_amd.sampler import export_u8
    from bndm_amd.vae import AutoencoderKL, vae_decode
    torch.set_num_threads(32)
    B, N = 8, 4
    m, U, cfg, sd = _pair(64, 4, 8, latent=True)
    z = torch.randn(B, 4, 64, 64, generator=torch.Generator().manual_seed(13))
    ones = np.ones(B, np.float32)
    ref_x0 = NO.get_noise_v2(z.numpy(), formula_L, ones, "GBN", "test")[0]
    L = torch.from_numpy(formula_L).cuda()
    x0 = get_noise_v2(torch.device("cuda"), z.cuda(), L, torch.ones(B).cuda(), None, "GBN", "test", True)[0]
    assert np.abs(x0.cpu().numpy() - ref_x0).max() <= 1e-4 * max(1.0, np.abs(ref_x0).max())
    xr = torch.from_numpy(np.ascontiguousarray(ref_x0))
    om = U.OracleUNet(cfg, sd)
    for t in range(N - 1, -1, -1):                                            # latent_iadb_bn_diffusers.py:524-529
        out = om(xr, torch.tensor((t + 1) / N))[0]
        xr = S.iadb_scheduler_step(out, t, xr, N, "gaussianBN", 8)
    sch = IADBScheduler(noise_type="gaussianBN", out_channels=8)
    sch.set_timesteps(N)
    xg = sch.sample(m, x0.contiguous())
    r = _rel(xg.cpu(), xr)
    print(f"c5 chain latents after {N} steps: rel-L2 {r:.3e}")
    assert r <= 3e-3
    vcfg = V.make_config(block_out_channels=(128, 256, 512, 512), layers_per_block=2)
    vsd = V.init_params(vcfg, seed=4, perturb_norm=0.1)
    vae = AutoencoderKL(block_out_channels=(128, 256, 512, 512), layers_per_block=2)
    vae.load_state_dict(vsd)
    vae = vae.cuda()
    # the reference decodes 0.18215-scaled latents; rescale the synthetic ones into that range on both sides
    ref_img = S.export_u8(V.vae_decode(vsd, vcfg, 0.18215 * xr[:2]), "round")
    got_img = export_u8(vae_decode(vae, 0.18215 * xg[:2]), "round").cpu().numpy()
    assert got_img.shape == ref_img.shape == (2, 512, 512, 3)
    mse = np.mean((got_img.astype(np.float64) - ref_img.astype(np.float64)) ** 2)
    psnr = 10 * np.log10(255.0 ** 2 / max(mse, 1e-12))
    print(f"c5 chain decoded images: PSNR {psnr:.1f} dB")
    assert psnr >= 35.0


def test_wide_concat_at_16x16_falls_back_to_igemm():
    """block_out_channels (128, 512, 512) at 32 px: the 16x16 up-block concatenates 512 + 512 channels, beyond conv_t32's
    scale / shift table -- those layers must run on the implicit-GEMM convolution + GroupNorm instead of failing at
    bndm_unet_finalize (diffusers accepts any such configuration; google/ddpm-*-256 has a 1024-channel concat)."""
    from oracle import unet_oracle as U
    from bndm_amd.unet import UNet2DModel
    boc = (128, 512, 512)
    cfg = dict(in_channels=3, out_channels=3, block_out_channels=boc, down_attn=(False,) * 3, up_attn=(False,) * 3,
               layers_per_block=2)
    sd = U.init_params(cfg, seed=5, perturb_norm=0.1)
    m = UNet2DModel(in_channels=3, out_channels=3, block_out_channels=boc, down_block_types=("DownBlock2D",) * 3,
                    up_block_types=("UpBlock2D",) * 3)
    m.load_state_dict(sd)
    m = m.to("cuda").eval()
    x = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(1))
    t = torch.tensor([0.7, 0.2])
    ref = U.forward(sd, cfg, x, t)
    got = m(x.cuda(), t.cuda(), return_dict=False)[0].cpu()
    kinds = {(k, lab.split()[-1]) for k, lab, _ in _ops(m, 2, 32)}
    assert ("conv_igemm", "16x16") in kinds and ("conv_t32<TH=8>", "32x32") in kinds, sorted(kinds)
    assert _rel(got, ref) <= 2e-3
