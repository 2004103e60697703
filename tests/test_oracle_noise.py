"""Pin oracle/noise_oracle.py against golden vectors captured from the reference's get_noise_v2."""
import os

import numpy as np
import pytest

from oracle import noise_oracle as O
from tests.golden_cases import NOISE_CASES, STRIDE, case_inputs, noise_case_shape, reference_draw


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "noise_cases.npz"))


def _check(gold, key, arrs, tol):
    for tag, a in zip(("n", "bn", "wn"), arrs):
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert tuple(gold[f"{key}|{tag}|shape"]) == a.shape
        g = gold[f"{key}|{tag}"]
        sub = a.reshape(-1)[::STRIDE]
        scale = max(1.0, float(np.abs(g).max()))
        assert np.abs(sub - g).max() <= tol * scale, (key, tag)
        s, sa = gold[f"{key}|{tag}|sum"]
        assert abs(a.astype(np.float64).sum() - s) <= 1e-5 * sa + 1e-3
        assert abs(np.abs(a.astype(np.float64)).sum() - sa) <= 1e-5 * sa + 1e-3


@pytest.mark.parametrize("ci", range(len(NOISE_CASES)))
def test_formula_L_cases(gold, formula_L, ci):
    res, nt, inplace, tt = NOISE_CASES[ci]
    B, C = noise_case_shape(res)
    x, alpha = case_inputs(1000 + ci, B, C, res)
    z = None if inplace else reference_draw(ci, res, nt, B, C)
    out = O.get_noise_v2(x, formula_L, alpha, nt, tt, z=z)
    # white-noise outputs are pure data movement -> exact; products differ by BLAS summation order
    _check(gold, f"formula|{res}|{nt}|{int(inplace)}|{tt}", out, 2e-5)


@pytest.mark.parametrize("ci", [i for i, c in enumerate(NOISE_CASES) if c[0] == 128 and c[2]])
def test_identity_L_exposes_permutation(gold, ci):
    res, nt, inplace, tt = NOISE_CASES[ci]
    B, C = noise_case_shape(res)
    x, alpha = case_inputs(1000 + ci, B, C, res)
    out = O.get_noise_v2(x, np.eye(4096, dtype=np.float32), alpha, nt, tt)
    _check(gold, f"identity|{res}|{nt}|1|{tt}", out, 1e-6)
    if nt != "gaussian":
        assert not np.array_equal(out[1], out[2])      # bn and wn are NOT pixel-aligned


def test_unsupported_sizes_raise(formula_L):
    x = np.zeros((1, 3, 16, 16), np.float32)
    with pytest.raises(NotImplementedError):
        O.get_noise_v2(x, formula_L, np.zeros(1, np.float32), "gaussianBN")
    with pytest.raises(NotImplementedError):
        O.get_noise_v2(np.zeros((1, 3, 32, 32), np.float32), formula_L, np.zeros(1, np.float32), "gaussian")
    with pytest.raises(NotImplementedError):
        O.get_noise_v2(np.zeros((1, 3, 64, 64), np.float32), formula_L, np.zeros(1, np.float32), "uniform")
