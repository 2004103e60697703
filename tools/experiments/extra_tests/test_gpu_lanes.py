"""NOT collected by `pytest tests/` on purpose: written while GPU access was closed (round 4), never run.  Move into tests/ once
it has passed on a GPU:   python -m pytest tools/experiments/extra_tests -m gpu -q -s

Lanes (bndm_unet_set_lanes, include/bndm_hip.h): the in-engine sampling loops as 2 / 4 chains of launches on separate HIP streams.
Samples are independent, the tile shapes are the handle's (max_batch), so every result must be BIT-IDENTICAL to one lane."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _models(lanes, cout=6, res=64, dtype="f16", threads=False):
    from bndm_amd.sampler import get_model
    return [get_model(3, cout, res, dtype=dtype, seed=5, lanes=n, lane_threads=threads).cuda().eval() for n in lanes]


@pytest.mark.parametrize("B,lanes,threads", [(8, 2, False), (8, 4, False), (6, 2, False), (8, 2, True), (8, 4, True)])
def test_iadb_loop_lanes_bit_identical(B, lanes, threads):
    from bndm_amd.sampler import sample_iadb
    m1, mn = _models((1, lanes), threads=threads)
    x0 = torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(B)).cuda()
    p = torch.tensor([1000.0, 0.0, 3.0], device="cuda")
    a, xa, _ = sample_iadb(m1, x0, 6, "sigmoid", p, 6, "gaussianBN", "test", log_freq=2)
    b, xb, _ = sample_iadb(mn, x0, 6, "sigmoid", p, 6, "gaussianBN", "test", log_freq=2)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert len(xa) == len(xb) and len(xa) >= 2
    for u, v in zip(xa, xb):
        assert torch.equal(u, v)
    # the caller's stream sees the loop as one in-order piece of work: an op queued right behind it reads the final x
    c = sample_iadb(mn, x0, 6, "sigmoid", p, 6, "gaussianBN", "train") * 1.0
    assert torch.equal(c, a)


def test_batch_not_divisible_runs_as_one_chain():
    from bndm_amd.sampler import sample_iadb
    m1, m2 = _models((1, 2))
    x0 = torch.randn(5, 3, 64, 64, generator=torch.Generator().manual_seed(1)).cuda()
    p = torch.tensor([1000.0, 0.0, 3.0], device="cuda")
    assert torch.equal(sample_iadb(m1, x0, 3, "sigmoid", p, 6, "gaussianBN", "train"),
                       sample_iadb(m2, x0, 3, "sigmoid", p, 6, "gaussianBN", "train"))


def test_ddim_loop_lanes_bit_identical():
    from bndm_amd.schedulers import DDIMScheduler
    m1, m2 = _models((1, 2), cout=3)
    sch = DDIMScheduler(num_train_timesteps=1000, beta_schedule="linear")
    sch.set_timesteps(5)
    x = torch.randn(8, 3, 64, 64, generator=torch.Generator().manual_seed(2)).cuda()
    assert torch.equal(sch.sample(m1, x.clone()), sch.sample(m2, x.clone()))


def test_conditional_loop_lanes_bit_identical():
    from bndm_amd.sampler import get_model, sample_iadb_conditional
    m1 = get_model(6, 6, 64, seed=7).cuda().eval()
    m2 = get_model(6, 6, 64, seed=7, lanes=2).cuda().eval()
    g = torch.Generator().manual_seed(4)
    x0, xc = torch.randn(4, 3, 64, 64, generator=g).cuda(), torch.randn(4, 3, 64, 64, generator=g).cuda()
    p = torch.tensor([1000.0, 0.0, 3.0], device="cuda")
    assert torch.equal(sample_iadb_conditional(m1, x0, xc, 3, "sigmoid", p, 6, "gaussianBN", "train"),
                       sample_iadb_conditional(m2, x0, xc, 3, "sigmoid", p, 6, "gaussianBN", "train"))


def test_benched_batch_two_lanes_and_forward_still_one_lane():
    """B=64 (the benchmarked size), 3 steps; a plain forward on a laned handle uses lane 0 and equals the one-lane forward"""
    from bndm_amd.sampler import sample_iadb
    m1, m2 = _models((1, 2))
    x0 = torch.randn(64, 3, 64, 64, generator=torch.Generator().manual_seed(9)).cuda()
    p = torch.tensor([1000.0, 0.0, 3.0], device="cuda")
    assert torch.equal(sample_iadb(m1, x0, 3, "sigmoid", p, 6, "gaussianBN", "train"),
                       sample_iadb(m2, x0, 3, "sigmoid", p, 6, "gaussianBN", "train"))
    t = torch.linspace(0.1, 0.9, 64).cuda()
    assert torch.equal(m1(x0, t, return_dict=False)[0], m2(x0, t, return_dict=False)[0])


def test_no_stagger_bit_identical():
    from bndm_amd.sampler import get_model, sample_iadb
    m1 = get_model(3, 6, 64, seed=5).cuda().eval()
    m2 = get_model(3, 6, 64, seed=5, lanes=4, lane_stagger=False).cuda().eval()
    x0 = torch.randn(8, 3, 64, 64, generator=torch.Generator().manual_seed(11)).cuda()
    p = torch.tensor([1000.0, 0.0, 3.0], device="cuda")
    assert torch.equal(sample_iadb(m1, x0, 4, "sigmoid", p, 6, "gaussianBN", "train"),
                       sample_iadb(m2, x0, 4, "sigmoid", p, 6, "gaussianBN", "train"))


def test_cu_share_lanes_bit_identical():
    from bndm_amd.sampler import get_model, sample_iadb
    m1 = get_model(3, 6, 64, seed=5).cuda().eval()
    m2 = get_model(3, 6, 64, seed=5, lanes=2, lane_cus=True).cuda().eval()
    x0 = torch.randn(8, 3, 64, 64, generator=torch.Generator().manual_seed(8)).cuda()
    p = torch.tensor([1000.0, 0.0, 3.0], device="cuda")
    assert torch.equal(sample_iadb(m1, x0, 4, "sigmoid", p, 6, "gaussianBN", "train"),
                       sample_iadb(m2, x0, 4, "sigmoid", p, 6, "gaussianBN", "train"))


def test_set_lanes_contract():
    import ctypes as C
    from bndm_amd import _lib
    lib = _lib.load()
    m = _models((2,))[0]
    h = m._ensure_engine(2, 64, torch.device("cuda", 0))
    assert lib.bndm_unet_set_lanes(h, 2, 0) == -2                      # BNDM_E_STATE: already finalised
    assert b"finalised" in lib.bndm_last_error()
    with pytest.raises(NotImplementedError):                        # BNDM_E_ARG: 1..4 lanes
        _models((5,))[0]._ensure_engine(2, 64, torch.device("cuda", 0))
