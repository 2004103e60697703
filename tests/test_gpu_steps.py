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
