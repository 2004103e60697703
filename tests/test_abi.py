"""CPU checks of the boundary: the library loads and exports every symbol include/bndm_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bndm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bndm_[a-z0-9_]+)\s*\(", text)))


def test_header_and_ctypes_table_agree():
    from bndm_amd import _lib
    assert _declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from bndm_amd import _lib
    lib = _lib.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.bndm_abi_version() == 1
    assert lib.bndm_bluenoise_workspace_bytes(64, 3, 64) == 4 * 192 * 4096 * 4
    assert lib.bndm_bluenoise_workspace_bytes(32, 3, 128) == 4 * 384 * 4096 * 4


def test_product_never_imports_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "bndm_amd")):
        for f in files:
            if f.endswith(".py") and re.search(r"^\s*(from|import)\s+oracle", open(os.path.join(base, f)).read(), re.M):
                bad.append(f)
    for f in ("utils.py", "iadb_bn.py", "ddim_diffusers.py", "latent_iadb_bn_diffusers.py",
              os.path.join("bluenoise", "get_noise_recent.py")):
        p = os.path.join(ROOT, f)
        if os.path.exists(p) and re.search(r"^\s*(from|import)\s+oracle", open(p).read(), re.M):
            bad.append(f)
    assert not bad, bad


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bndm_amd import _lib
    from bluenoise.get_noise_recent import get_noise_v2
    x = torch.zeros(1, 3, 64, 64)
    with pytest.raises(_lib.BndmError):
        get_noise_v2(torch.device("cpu"), x, torch.eye(4096), torch.zeros(1), None, "gaussianBN", "test", True)


def test_product_library_has_no_ablation_kernels_and_few_switches():
    """The wrong-result profiling variants of conv_t32 (template argument ABL != 0) are compiled only into
    tools/libbndm_ablate.so (-DBNDM_ABLATION); the product library holds ABL = 0 instantiations only, and its sources
    read at most eight environment switches."""
    from bndm_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    names = set(re.findall(rb"conv_t32IDF16b?_?Li(\d+)ELi(\d+)E", blob))
    assert names, "conv_t32 kernels not found in the library"
    assert {abl for _, abl in names} == {b"0"}, sorted(names)
    n = 0
    csrc = os.path.join(ROOT, "bndm_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith(".hip"):
            n += open(os.path.join(csrc, f)).read().count("getenv")
    assert n <= 8, n


def test_bench_refuses_the_ablation_switch_and_reports_switches():
    import subprocess
    import sys
    env = dict(os.environ, BNDM_ABLATE="15")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-cpu"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "BNDM_ABLATE" in (r.stderr + r.stdout)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"env": env_seen' in src and '"other_configs": others' in src
