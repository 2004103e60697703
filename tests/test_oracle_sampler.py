"""Pin oracle/sampler_oracle.py against goldens captured from the reference's utils.py."""
import os

import numpy as np
import pytest
import torch

from oracle import sampler_oracle as S
from tests.golden_cases import LOOP_CASES, FakeModel, case_inputs


@pytest.fixture(scope="module")
def sched(golden_dir):
    return np.load(os.path.join(golden_dir, "schedules.npz"))


@pytest.fixture(scope="module")
def loops(golden_dir):
    return np.load(os.path.join(golden_dir, "loops.npz"))


def test_schedules_bit_exact(sched):
    for key in sched.files:
        parts = key.split("|")
        N = int(parts[2])
        t = torch.arange(0, N + 1).float()
        if parts[0] == "alpha":
            got = S.alpha_schedule(t, N, "linear")
        elif parts[1] == "linear":
            got = S.gamma_schedule(t, N, "linear", (1.0, 0.0, 3.0))
        else:
            params = torch.tensor(eval(parts[3]))
            got = S.gamma_schedule(t, N, parts[1], params)
        assert np.array_equal(got.numpy(), sched[key]), key


def test_gamma_endpoints(sched):
    for key in sched.files:
        if key.startswith("gamma|sigmoid"):
            g = sched[key]
            assert g[0] == 0.0 and g[-1] == 1.0


@pytest.mark.parametrize("ci", range(len(LOOP_CASES)))
def test_iadb_loop_matches_reference(loops, ci):
    nt, oc, gs, params, N = LOOP_CASES[ci]
    x0, _ = case_inputs(2000 + ci, 2, 3, 8)
    x, snaps = S.sample_iadb(FakeModel(oc), torch.from_numpy(x0.copy()), N, gs, torch.tensor(params),
                             oc, nt, "test")
    key = f"{nt}|{oc}|{gs}|{params}|{N}"
    assert np.array_equal(x.numpy(), loops[key + "|final"])
    assert len(snaps) == int(loops[key + "|nsnap"])
    assert np.array_equal(snaps[0].numpy(), loops[key + "|snap_first"])
    assert np.array_equal(snaps[len(snaps) // 2].numpy(), loops[key + "|snap_mid"])


def test_iadb_scheduler_step_known_answers():
    x = torch.full((1, 4, 2, 2), 1.0)
    d = torch.cat([torch.full((1, 4, 2, 2), 2.0), torch.full((1, 4, 2, 2), -4.0)], 1)
    y = S.iadb_scheduler_step(d, 3, x, 10, "gaussianBN", 8)
    assert torch.allclose(y, torch.full_like(x, 1.0 + 0.1 * 2.0 - 0.1 * 4.0))
    y = S.iadb_scheduler_step(d[:, :4], 3, x, 10, "gaussian", 4)
    assert torch.allclose(y, torch.full_like(x, 1.2))


def test_ddim_tables_and_step_known_answers():
    acp, ts, ratio = S.ddim_tables(num_inference=100)
    assert ratio == 10 and ts[0] == 990 and ts[-1] == 0 and len(ts) == 100
    # eps = 0, x inside the clip range: x' = sqrt(a_prev/a_t) x
    x = torch.full((1, 3, 2, 2), 0.01)
    y = S.ddim_step(torch.zeros_like(x), 500, x, acp, ratio)
    assert torch.allclose(y, x * (acp[490] / acp[500]) ** 0.5)
    # last step: alpha_prev = 1 -> returns clipped x0 prediction
    y = S.ddim_step(torch.zeros_like(x), 0, x, acp, ratio)
    assert torch.allclose(y, (x / acp[0] ** 0.5).clamp(-1, 1))


def test_export_u8():
    x = torch.tensor([-1.5, -1.0, 0.0, 0.999, 1.0, 2.0]).view(1, 1, 1, 6).repeat(1, 3, 1, 1)
    assert S.export_u8(x, "trunc")[0, 0, :, 0].tolist() == [0, 0, 127, 254, 255, 255]
    assert S.export_u8(x, "round")[0, 0, :, 0].tolist() == [0, 0, 128, 255, 255, 255]


def test_train_targets_known_answers():
    """iadb_bn.py:915,946-956: hand-evaluated blend and targets (x1 data, x0 noise)."""
    x0 = torch.full((2, 1, 2, 2), 2.0)
    x1 = torch.full((2, 1, 2, 2), -1.0)
    bn = torch.full((2, 1, 2, 2), 0.5)
    wn = torch.full((2, 1, 2, 2), 0.25)
    alpha = torch.tensor([0.25, 1.0])
    alpha_prev = torch.tensor([0.0, 0.5])
    xa, t1, t2, t = S.train_targets(x0, x1, bn, wn, alpha, alpha_prev)
    assert torch.equal(xa[0], torch.full((1, 2, 2), 0.25 * 2.0 + 0.75 * -1.0))
    assert torch.equal(xa[1], torch.full((1, 2, 2), 2.0))               # alpha = 1: pure noise
    assert torch.equal(t1, torch.full((2, 1, 2, 2), -3.0))
    assert torch.equal(t2[0], torch.zeros(1, 2, 2)) and torch.equal(t2[1], torch.full((1, 2, 2), 0.125))
    assert torch.equal(t, t1 + t2)
    xa2, t1b, t2b, tb = S.train_targets(x0, x1, None, None, alpha, None)
    assert t2b is None and torch.equal(tb, t1b) and torch.equal(xa2, xa)


def test_train_targets_is_the_forward_of_the_sampler_step():
    """Consistency the reference relies on (iadb_bn.py:902-911): stepping x_alpha(t) by the targets with
    d_alpha = alpha_t - alpha_{t-1}, d_gamma = gamma_t - gamma_{t-1} lands on x_alpha(t-1) when x0 is the
    gamma-blend of one blue and one white draw."""
    g = torch.Generator().manual_seed(5)
    B = 3
    x1 = torch.randn(B, 3, 8, 8, generator=g, dtype=torch.float64)
    bn = torch.randn(B, 3, 8, 8, generator=g, dtype=torch.float64)
    wn = torch.randn(B, 3, 8, 8, generator=g, dtype=torch.float64)
    a_t, a_p = torch.tensor([0.8, 0.5, 0.2], dtype=torch.float64), torch.tensor([0.7, 0.4, 0.1], dtype=torch.float64)
    g_t, g_p = torch.tensor([0.9, 0.6, 0.3], dtype=torch.float64), torch.tensor([0.85, 0.45, 0.05], dtype=torch.float64)
    v = lambda s: s.view(-1, 1, 1, 1)
    x0_t = bn * (1 - v(g_t)) + wn * v(g_t)            # get_noise_recent.py:116
    x0_p = bn * (1 - v(g_p)) + wn * v(g_p)
    xa_t, tar1, tar2, _ = S.train_targets(x0_t, x1, bn, wn, a_t, a_p)
    xa_p = v(a_p) * x0_p + (1 - v(a_p)) * x1
    # exact identity: x_alpha(t-1) = x_alpha(t) + d_alpha*(x1 - x0_t) + d_gamma*alpha_{t-1}*(bn - wn)
    stepped = xa_t + v(a_t - a_p) * tar1 + v(g_t - g_p) * tar2
    assert torch.allclose(stepped, xa_p, atol=1e-12)
