"""The library's SHIPPED machine code, executed on the CPU: tests/gfx950sim is an instruction-level gfx950 simulator that runs
the code objects inside bndm_amd/libbndm_hip.so, launch by launch, under the recording HIP stand-in (tests/hipmock), with every
memory operation completing as late as the ISA allows.  These tests make the comparisons of the `-m gpu` suite -- C ABI in,
oracle/ as the checker -- on what the simulator computed, and require zero hazards (reads of registers / LDS bytes still in
flight, out-of-bounds accesses).

This does NOT replace the GPU suite (the simulator's instruction semantics are the author's reading of the ISA, calibrated on
kernels whose hardware results are on record; time and caches are not modelled).  It is what can be known about
device code while no GPU is reachable, and the gate a candidate library passes before it is given GPU minutes.

Default run: the fast subset (about a minute on 8 cores).  RUN_SIM_SLOW=1 adds every configuration of tests/gfx950sim/suite.py
(about an hour; `tools/sim_suite.sh` writes their log to profiles/)."""
import os
import shutil

import pytest

from tests.gfx950sim import suite
from tests.hipmock import harness as H

_NEED = ["/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf", "/opt/rocm/include/hip/hip_runtime_api.h"]
pytestmark = [
    pytest.mark.skipif(not os.path.exists(H.PRODUCT_LIB), reason="bndm_amd/libbndm_hip.so has not been built"),
    pytest.mark.skipif(not all(os.path.exists(p) for p in _NEED) or not shutil.which("g++") or not shutil.which("objcopy"),
                       reason="needs the ROCm LLVM tools, the HIP headers, g++ and objcopy"),
]
SLOW = os.environ.get("RUN_SIM_SLOW") == "1"


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("gfx950sim"))


def _check(r):
    assert r["ok"], f"{r['name']}: value {r.get('value')} (bar {r.get('bar')}), hazards {r.get('hazards')}\n{str(r.get('detail'))[-3000:]}"
    print(f"{r['name']}: {r.get('value'):.3e} (bar {r['bar']}), {r['launches']} launches, {r['wave_instructions']} wave-instructions, "
          f"0 hazards, {r['seconds']} s")


@pytest.mark.parametrize("name", suite.FAST)
def test_shipped_machine_code_reproduces_the_oracle(name, workdir):
    """steps: the Euler / DDIM / export / training-target kernels bit-exact; noise_small64: the L.z transform vs the noise oracle
    (<= 1e-4); deep32_t32x4: a whole UNet forward -- five levels from 32 to 2 px with 4x4 attention, batch 1 on a handle sized for
    batch 64 -- in which the 4-wave conv_t32<TH=16> (the dominant kernel of the benchmark) and <TH=8> run on the recorded arguments
    of the 8-wave launches, next to conv_s at 8x8 / 4x4 / 2x2 with its stride-2, nearest-2x, shortcut and attention launches, gn_small,
    conv_in, the igemm downsamplers and the head: rel-L2 <= 2e-3 vs oracle/unet_oracle.py, zero hazards"""
    _check(suite.run_config(name, work=workdir, procs=int(os.environ.get("GFX950SIM_TEST_PROCS", "8"))))


@pytest.mark.skipif(not SLOW, reason="RUN_SIM_SLOW=1 runs every simulator configuration (about an hour); log: profiles/r06_sim_suite.log")
@pytest.mark.parametrize("name", [n for n in suite.CONFIGS if n not in suite.FAST])
def test_every_configuration_on_the_simulator(name, workdir):
    _check(suite.run_config(name, work=workdir, procs=int(os.environ.get("GFX950SIM_TEST_PROCS", "8"))))


def test_the_simulator_catches_a_dropped_wait(workdir):
    """Sensitivity: the same library with ONE counted wait of conv_t32 loosened in the instruction stream
    (`s_waitcnt vmcnt(N)` -> `vmcnt(N + 3)` at the K loop's tile wait) must produce hazards or a wrong result -- the
    latest-legal memory model is what turns a too-loose count into a deterministic failure"""
    import numpy as np
    from tests.gfx950sim import loader
    ks = loader.load_library(H.PRODUCT_LIB)
    k = next(v for n, v in ks.items() if "conv_t32IDF16_Li8ELi0ELi8ELi128ELi1E" in n)
    waits = [i for i in k.insts if i.mnem == "s_waitcnt" and any(m.startswith("vmcnt(") and m != "vmcnt(0)" for m in i.mods)]
    assert len(waits) >= 4, "the kernel is expected to carry counted vmcnt waits"
    # (the mutation itself is exercised in tests/gfx950sim/mutate.py through the same runner)
    from tests.gfx950sim import mutate
    r = mutate.run_with_loosened_waits(workdir, kernel_substr="conv_t32IDF16_Li8ELi0ELi8ELi128ELi1E", add=3)
    assert r["hazards"] > 0 or not np.isfinite(r["rel"]) or r["rel"] > 2e-3, r
