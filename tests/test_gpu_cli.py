"""End-to-end runs of the drop-in CLIs on the GPU with tiny workloads (synthetic factor / weights)."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def workdir(tmp_path):
    cwd = os.getcwd()
    yield str(tmp_path)
    os.chdir(cwd)


def test_iadb_cli_full_batches_and_replicability(workdir):
    from bndm_amd.cli_iadb import main
    argv = ("--dataset=cat_res64 --res=64 --batch_size=3 --train_or_test=test --nb_steps=4 --test_samples=5 "
            "--noise_type=gaussianBN --scheduler_gamma=sigmoid --scheduler_param=1000 --out_channel=6 "
            f"--full_batches --save_noise --root={workdir}").split()
    assert main(argv) == 0
    run = os.path.join(workdir, "results_gaussianBN", "cat_res64_gaussianBN_sigmoid_1000.0_0_3_outc6_seed0",
                       "cat_res64_iadb_gwn2gbn_steps4")
    assert len(glob.glob(os.path.join(run, "images", "*.png"))) == 5          # 3 + 2 (ragged last batch)
    assert os.path.exists(os.path.join(run, "images", "00005.png"))
    npz = sorted(glob.glob(os.path.join(run, "noise", "noise_batch*_idx*.npz")))
    assert sorted(np.load(f)["noise"].shape for f in npz) == [(2, 3, 64, 64), (3, 3, 64, 64)]
    assert glob.glob(os.path.join(run, "seqs", "gwn2gbn_img*_step*.png"))
    # reference behaviour (replicability clamps): cat_res64 only runs batch id 4 and a single sample
    argv = ("--dataset=cat_res64 --res=64 --batch_size=2 --train_or_test=test --nb_steps=3 --test_samples=12 "
            f"--noise_type=gaussian --scheduler_gamma=linear --scheduler_param=1 --out_channel=3 --root={workdir}").split()
    assert main(argv) == 0
    run = os.path.join(workdir, "results_gaussianBN", "cat_res64_gaussian_linear_outc3_seed0", "cat_res64_iadb_gwn_steps3")
    assert len(glob.glob(os.path.join(run, "images", "*.png"))) == 1


def test_ddim_latent_and_superres_cli(workdir):
    from bndm_amd.cli_ddim import main as ddim
    from bndm_amd.cli_iadb import main as iadb
    from bndm_amd.cli_latent import main as latent
    assert ddim(("--dataset_name=church_res64 --train_or_test=test --eval_batch_size=2 --test_samples=2 --resolution=64 "
                 f"--output_dir=ddim_church_res64 --ddpm_num_inference_steps=5 --full_batches --root={workdir}").split()) == 0
    assert len(glob.glob(os.path.join(workdir, "results_gaussianBN", "ddim_church_res64", "images", "ddim_img*.png"))) == 2
    os.chdir(os.path.dirname(workdir))
    assert latent(("--dataset_name=cat_res512 --resolution=512 --train_or_test=test --eval_batch_size=2 --test_samples=2 "
                   "--output_dir=latent_iadb_cat_res512 --out_channels=4 --noise_type=gaussianBN "
                   f"--ddpm_num_inference_steps=3 --full_batches --root={workdir}").split()) == 0
    lat = glob.glob(os.path.join(workdir, "results_gaussianBN", "latent_iadb_cat_res512_gaussianBN", "latents", "*.npy"))
    assert len(lat) == 2 and np.load(lat[0]).shape == (4, 64, 64)
    pngs = sorted(glob.glob(os.path.join(workdir, "results_gaussianBN", "latent_iadb_cat_res512_gaussianBN", "images",
                                         "iadb_gwn2gbn_*.png")))
    assert [os.path.basename(p) for p in pngs] == ["iadb_gwn2gbn_00001.png", "iadb_gwn2gbn_00002.png"]   # :566
    from PIL import Image
    assert Image.open(pngs[0]).size == (512, 512)                      # decoded by the HIP AutoencoderKL decoder
    os.chdir(os.path.dirname(workdir))
    assert iadb(("--dataset=church_res128 --res=128 --batch_size=1 --train_or_test=test --nb_steps=3 --test_samples=2 "
                 "--is_conditional --noise_type=gaussianBN --scheduler_gamma=sigmoid --scheduler_param=0.2 "
                 f"--out_channel=6 --conditional_type=superres --root={workdir}").split()) == 0
    imgs = glob.glob(os.path.join(workdir, "results_gaussianBN_superres", "*", "*", "images", "*.png"))
    assert len(imgs) >= 1
