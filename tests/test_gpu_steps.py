"""GPU parity of the sampler-step kernels vs the oracle (bit-exact: same fp32 operation order)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("B,Cc,Cout,HW", [(4, 3, 3, 4096), (4, 3, 6, 4096), (3, 4, 8, 1024), (2, 3, 6, 64)])
def test_iadb_step_bit_exact(B, Cc, Cout, HW):
    from bndm_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 100 + Cout)
    x = torch.randn(B, Cc, HW, generator=g)
    d = torch.randn(B, Cout, HW, generator=g)
    da, dg = np.float32(1 / 250), np.float32(0.0040531)
    ref = x + torch.tensor(da) * d[:, :Cc]
    if Cout == 2 * Cc:
        ref = ref + torch.tensor(dg) * d[:, Cc:]
    xd, dd = x.cuda(), d.cuda()
    _lib.check(lib.bndm_iadb_step(_p(xd), _p(dd), float(da), float(dg), B, Cc, Cout, HW, _lib.current_stream_ptr()),
               "iadb_step")
    assert torch.equal(xd.cpu(), ref)


def test_iadb_step_rejects_bad_out_channel():
    from bndm_amd import _lib
    lib = _lib.load()
    x = torch.zeros(1, 3, 64, device="cuda")
    d = torch.zeros(1, 5, 64, device="cuda")
    with pytest.raises(NotImplementedError):
        _lib.check(lib.bndm_iadb_step(_p(x), _p(d), 0.1, 0.1, 1, 3, 5, 64, _lib.current_stream_ptr()), "iadb_step")


def test_ddim_step_matches_oracle():
    from bndm_amd import _lib
    from oracle import sampler_oracle as S
    lib = _lib.load()
    acp, ts, ratio = S.ddim_tables(num_inference=100)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 64, 64, generator=g) * 1.5
    eps = torch.randn(2, 3, 64, 64, generator=g)
    for t in (990, 500, 0):
        ref = S.ddim_step(eps, t, x, acp, ratio)
        a_t = acp[t]
        a_p = acp[t - ratio] if t - ratio >= 0 else torch.tensor(1.0)
        xd = x.cuda().clone()
        _lib.check(lib.bndm_ddim_step(_p(xd), _p(eps.cuda()), float(a_t ** 0.5), float((1 - a_t) ** 0.5),
                                      float(a_p ** 0.5), float((1 - a_p) ** 0.5), 1.0, x.numel(),
                                      _lib.current_stream_ptr()), "ddim_step")
        assert (xd.cpu() - ref).abs().max().item() <= 1e-6 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("rounding", [0, 1])
def test_export_u8_bit_exact(rounding):
    from bndm_amd import _lib
    from oracle import sampler_oracle as S
    lib = _lib.load()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(3, 3, 32, 32, generator=g) * 0.8
    x[0, 0, 0, :6] = torch.tensor([-1.5, -1.0, 0.0, 0.999, 1.0, 2.0])
    x[0, 1, 0, :4] = torch.tensor([1 / 255.0 - 1, 3 / 255.0 - 1, 0.00392157, -0.00392157])
    ref = S.export_u8(x, "trunc" if rounding == 0 else "round")
    out = torch.empty(3, 32, 32, 3, dtype=torch.uint8, device="cuda")
    _lib.check(lib.bndm_export_u8(_p(x.cuda()), _p(out), 3, 3, 1024, rounding, _lib.current_stream_ptr()), "export")
    assert np.array_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("two", [True, False])
def test_train_targets_bit_exact(two):
    """bndm_iadb_train_targets vs the oracle's torch fp32 operation order (iadb_bn.py:915,946-956)."""
    from bndm_amd.training import train_targets
    from oracle import sampler_oracle as S
    g = torch.Generator().manual_seed(11)
    B = 5
    x0 = torch.randn(B, 3, 64, 64, generator=g)
    x1 = torch.randn(B, 3, 64, 64, generator=g)
    bn = torch.randn(B, 3, 64, 64, generator=g) if two else None
    wn = torch.randn(B, 3, 64, 64, generator=g) if two else None
    alpha = torch.rand(B, generator=g)
    alpha_prev = torch.rand(B, generator=g) if two else None
    ref = S.train_targets(x0, x1, bn, wn, alpha, alpha_prev)
    dev = lambda t: None if t is None else t.cuda()
    got = train_targets(dev(x0), dev(x1), dev(bn), dev(wn), dev(alpha), dev(alpha_prev))
    for r, o in zip(ref, got):
        assert (r is None) == (o is None)
        if r is not None:
            assert torch.equal(o.cpu(), r)


def test_noise_injection_matches_oracle_pipeline():
    """get_noise_v2('train', inplace=False) + blend + targets on the HIP path vs oracle noise + oracle targets on
    the same white draw (the draw is replayed through global_z)."""
    from bndm_amd.training import train_targets
    from bndm_amd.bluenoise import get_noise_v2
    from bndm_amd.synth import formula_factor
    from oracle import noise_oracle as NO
    from oracle import sampler_oracle as S
    L = torch.from_numpy(formula_factor())
    B = 3
    g = torch.Generator().manual_seed(3)
    x1 = torch.randn(B, 3, 64, 64, generator=g)
    z = torch.randn(B, 3, 64, 64, generator=g)
    gamma_t = torch.tensor([0.2, 0.5, 0.9])
    alpha, alpha_prev = torch.tensor([0.3, 0.6, 1.0]), torch.tensor([0.29, 0.59, 0.99])
    x0, bn, wn = get_noise_v2(torch.device("cuda"), x1.cuda(), L.cuda(), gamma_t.cuda(), None, "gaussianBN", "train",
                              False, global_z=z.cuda())
    rx0, rbn, rwn = NO.get_noise_v2(x1.numpy(), L.numpy(), gamma_t.numpy(), "gaussianBN", "train", z=z.numpy())
    assert np.abs(x0.cpu().numpy() - rx0).max() <= 1e-4 * np.abs(rx0).max()
    got = train_targets(x0, x1.cuda(), bn, wn, alpha.cuda(), alpha_prev.cuda())
    ref = S.train_targets(x0.cpu(), x1, bn.cpu(), wn.cpu(), alpha, alpha_prev)
    for r, o in zip(ref, got):
        assert torch.equal(o.cpu(), r)
