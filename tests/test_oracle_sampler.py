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
