"""CLI surface: every flag set used by the reference's scripts/sampling/*.sh parses (CPU only)."""
import pytest

IADB_LINES = [
    "--dataset=cat_res128 --res=128 --batch_size=200 --train_or_test=test --nb_steps=250 --test_samples=30000 "
    "--noise_type=gaussian --scheduler_gamma=linear --scheduler_param=1 --out_channel=3",
    "--dataset=cat_res64 --res=64 --batch_size=500 --train_or_test=test --nb_steps=250 --test_samples=30000 "
    "--noise_type=gaussianBN --scheduler_gamma=sigmoid --scheduler_param=1000 --out_channel=6",
    "--dataset=celeba_res64 --res=64 --batch_size=500 --train_or_test=test --nb_steps=250 --test_samples=30000 "
    "--noise_type=gaussianBN --scheduler_gamma=linear --scheduler_param=1 --out_channel=3",
    "--dataset=church_res128 --res=128 --batch_size=200 --train_or_test=test --nb_steps=250 --test_samples=100 "
    "--is_conditional --noise_type=gaussianBN --scheduler_gamma=sigmoid --scheduler_param=0.2 --out_channel=6 "
    "--conditional_type=superres",
]
DDIM_LINE = ("--dataset_name=church_res64 --train_or_test=test --eval_batch_size=500 --test_samples=30000 "
             "--resolution=64 --random_flip --output_dir=ddim_church_res64 --train_batch_size=2 --num_epochs=1000 "
             "--gradient_accumulation_steps=1 --learning_rate=1e-4 --lr_warmup_steps=0")
LATENT_LINE = ("--dataset_name=cat_res512 --resolution=512 --train_or_test=test --eval_batch_size=50 "
               "--test_samples=100 --random_flip --output_dir=latent_iadb_cat_res512 --train_batch_size=256 "
               "--num_epochs=1000 --gradient_accumulation_steps=1 --learning_rate=1e-4 --lr_warmup_steps=0 "
               "--out_channels=4 --noise_type=gaussianBN")


@pytest.mark.parametrize("line", IADB_LINES)
def test_iadb_flags_parse(line):
    from bndm_amd.cli_iadb import build_parser
    opt = build_parser().parse_args(line.split())
    assert opt.train_or_test == "test" and opt.nb_steps == 250


def test_output_folder_naming_matches_reference():
    # iadb_bn.py:486-496: linear -> no schedule params in the name; otherwise tau_s_e with argparse's types
    from bndm_amd.cli_iadb import build_parser, output_folder
    o = build_parser().parse_args(IADB_LINES[1].split())
    assert output_folder(o) == "results_gaussianBN/cat_res64_gaussianBN_sigmoid_1000.0_0_3_outc6_seed0"
    o = build_parser().parse_args(IADB_LINES[0].split())
    assert output_folder(o) == "results_gaussianBN/cat_res128_gaussian_linear_outc3_seed0"
    o = build_parser().parse_args(IADB_LINES[3].split())
    assert output_folder(o).startswith("results_gaussianBN_superres/church_res128_gaussianBN_sigmoid_0.2_0_3_outc6")


def test_ddim_and_latent_flags_parse():
    from bndm_amd.cli_ddim import build_parser as ddim
    from bndm_amd.cli_latent import build_parser as latent
    import input_args
    d = ddim().parse_args(DDIM_LINE.split())
    assert d.resolution == 64 and d.ddpm_num_inference_steps == 250 and d.eval_batch_size == 500
    l = latent().parse_args(LATENT_LINE.split())
    assert l.noise_type == "gaussianBN" and l.out_channels == 4
    assert input_args.parse_args(LATENT_LINE.split()).resolution == 512


def test_schedule_tables_match_goldens(golden_dir):
    """Host step tables (K6) reproduce the reference's per-step alpha/gamma differences: alpha exactly,
    gamma to 2 ulp of gamma (torch's vectorised sigmoid rounds the same argument differently in the SIMD
    body and in the scalar tail of a tensor, so even the reference's own bits depend on the batch size)."""
    import os
    import numpy as np
    from bndm_amd.schedules import step_tables
    g = np.load(os.path.join(golden_dir, "schedules.npz"))
    for N in (250, 1000, 100, 50):
        for key, kind, params in ((f"gamma|sigmoid|{N}|(1000.0, 0.0, 3.0)", "sigmoid", (1000.0, 0.0, 3.0)),
                                  (f"gamma|sigmoid|{N}|(0.2, 0.0, 3.0)", "sigmoid", (0.2, 0.0, 3.0)),
                                  (f"gamma|linear|{N}", "linear", (1.0, 0.0, 3.0))):
            t_in, da, dg = step_tables(N, "linear", kind, params)
            alpha, gamma = g[f"alpha|linear|{N}"], g[key]
            assert np.array_equal(t_in, alpha[1:][::-1])
            assert np.array_equal(da, (alpha[1:] - alpha[:-1])[::-1])
            assert np.abs(dg - (gamma[1:] - gamma[:-1])[::-1]).max() <= 2.5e-7


def test_host_schedules_match_goldens(golden_dir):
    import os
    import numpy as np
    import torch
    from utils import get_scheduler, get_scheduler_gamma
    g = np.load(os.path.join(golden_dir, "schedules.npz"))
    for key in g.files:
        parts = key.split("|")
        N = int(parts[2])
        t = torch.arange(0, N + 1).float()
        if parts[0] == "alpha":
            got = get_scheduler(t, "linear", N)
        elif parts[1] == "linear":
            got = get_scheduler_gamma(t, "linear", [1.0, 0.0, 3.0], N)
        else:
            got = get_scheduler_gamma(t, parts[1], torch.tensor(eval(parts[3])), N)
        assert np.array_equal(got.numpy(), g[key]), key


def test_metrics_ssim_psnr_properties():
    """bndm_amd/metrics.py (restatement of the piq calls at iadb_bn.py:636-644): identities and monotonicity."""
    import torch
    from bndm_amd.metrics import psnr, ssim
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 3, 64, 64, generator=g)
    assert torch.allclose(ssim(x, x), torch.ones(2, dtype=torch.float64))
    n1 = (x + 0.05 * torch.randn(x.shape, generator=g)).clamp(0, 1)
    n2 = (x + 0.20 * torch.randn(x.shape, generator=g)).clamp(0, 1)
    s1, s2 = ssim(x, n1), ssim(x, n2)
    assert (s1 > s2).all() and (s2 > 0).all() and (s1 < 1).all()
    assert torch.allclose(ssim(x, n1), ssim(n1, x))
    assert (psnr(x, n1) > psnr(x, n2)).all()
    flat = torch.full((1, 1, 32, 32), 0.5)
    assert abs(float(psnr(flat, flat + 0.1)) - 20.0) < 1e-4           # mse = 0.01 -> 20 dB
    assert ssim(torch.rand(1, 3, 512, 512, generator=g), torch.rand(1, 3, 512, 512, generator=g)).shape == (1,)
