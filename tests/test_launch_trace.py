"""Host side of libbndm_hip.so without a GPU: the library runs against a recording stand-in for the HIP runtime
(tests/hipmock: kernels are recorded, not executed here) and its trace -- launch list, grids, LDS sizes, kernel-argument bytes,
table uploads, buffer layout -- is compared with the digest of the PINNED build (tests/golden/launch_traces.json).  What backs
the pinned build is written in the fixture ("validated_by": a committed GPU-suite log naming its sha256, or, while no GPU is
reachable, the simulator suite's log -- tests/gfx950sim executes the machine code on the CPU); the tests below check that the
named log exists and names that library.  What this pins: a later build asks the GPU for exactly the same work, in the same
order, on the same buffers, with the same machine code, as the pinned one."""
import json
import os

import shutil

import pytest

from tests.hipmock import harness as H

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "launch_traces.json")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# everything here runs the built library under a stand-in compiled on the spot and reads its code objects with the ROCm LLVM
# tools: skip (do not error) on a box without them
_NEED = ["/opt/rocm/lib/llvm/bin/llvm-readelf", "/opt/rocm/include/hip/hip_runtime_api.h"]
pytestmark = [
    pytest.mark.skipif(not os.path.exists(H.PRODUCT_LIB), reason="bndm_amd/libbndm_hip.so has not been built (__graft_entry__.build())"),
    pytest.mark.skipif(not all(os.path.exists(p) for p in _NEED) or not shutil.which("g++") or not shutil.which("objcopy"),
                       reason="needs the ROCm LLVM tools, the HIP headers, g++ and objcopy"),
]


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("hipmock"))


from tests.hipmock.exec_forward import CASES as _CASES     # (in, out, resolution, layout, batch, mode) per case
REPLAY_CASES = {k: (v[0], v[1], v[2], v[4], v[5]) for k, v in _CASES.items()}

# Every test below needs the output of a subprocess that runs under the stand-in (5 .. 25 s each, mostly single-threaded set-up).
# They are independent, so the first test that asks starts ALL of them on a small pool and each test waits for its own.
import concurrent.futures
import threading

_POOL = concurrent.futures.ThreadPoolExecutor(max_workers=2)
_THREADS = {k: "4" for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}      # two jobs x four BLAS threads = the 8 cores
_JOBS, _LOCK = {}, threading.Lock()
_WEIGHTS = {}


def oracle_weights(workdir, key, make):
    """state dict of the oracle's initialisation (the one the GPU parity tests use) + its .npz for the replay process, made once
    per network layout"""
    import numpy as np
    with _LOCK:
        if key not in _WEIGHTS:
            _WEIGHTS[key] = [threading.Lock(), None]
        ent = _WEIGHTS[key]
    with ent[0]:
        if ent[1] is None:
            sd = make()
            wfile = os.path.join(workdir, f"weights_{abs(hash(key))}.npz")
            np.savez(wfile, **{k: v.numpy() for k, v in sd.items()})
            ent[1] = (sd, wfile)
    return ent[1]


def _case_network(case):
    """-> (cfg, key, make) of a replay case's network"""
    from oracle import unet_oracle as UO
    from tests.hipmock.exec_forward import SIM_CASES
    cin, cout, res, layout, B, mode = {**_CASES, **SIM_CASES}[case]
    if mode == "vae":
        from oracle import vae_oracle as VO
        cfg = VO.make_config()
        return cfg, "vae", lambda: VO.init_params(cfg, seed=0, perturb_norm=0.1)
    boc, da_, ua_ = layout
    cfg = dict(in_channels=cin, out_channels=cout, block_out_channels=tuple(boc), layers_per_block=2,
               down_attn=tuple(i == da_ for i in range(len(boc))), up_attn=tuple(i == ua_ for i in range(len(boc))))
    if len(boc) >= 6 and res >= 64:
        assert cfg == UO.make_config(res, cin, cout)         # the reference's constructor arguments for this resolution
    return cfg, (cin, cout, tuple(boc), da_, ua_), lambda: UO.init_params(cfg, seed=0, perturb_norm=0.1)


def _replay_job(workdir, tag, case, env):
    cfg, key, make = _case_network(case)
    sd, wfile = oracle_weights(workdir, key, make)
    out_dir = os.path.join(workdir, tag)
    out = H.run_script("exec_forward.py", H.PRODUCT_LIB, out_dir, out_dir, case, wfile, env=dict(_THREADS, **(env or {})), mockdir=workdir)
    return cfg, sd, out_dir, out


def _start_all(workdir):
    with _LOCK:
        if _JOBS:
            return
        H.build_mock(workdir)
        H.kernargs_file(H.PRODUCT_LIB, workdir)               # (shared by every job: made before the pool starts)
        for case in _CASES:
            _JOBS[("replay", case)] = _POOL.submit(_replay_job, workdir, case, case, None)
        for switch in ("BNDM_NO_TAIL", "BNDM_NO_FUSED"):
            _JOBS[("replay", switch)] = _POOL.submit(_replay_job, workdir, switch, "c2", {switch: "1"})
        _JOBS[("replay", "mb64")] = _POOL.submit(_replay_job, workdir, "mb64", "c2", {"EXEC_MAX_BATCH": "64"})
        for script in ("check_conv_t32.py", "check_conv_s.py"):
            _JOBS[("script", script)] = _POOL.submit(H.run_script, script, H.PRODUCT_LIB, os.path.join(workdir, script), env=_THREADS, mockdir=workdir)
        for name in H.SCENARIOS:
            _JOBS[("trace", name)] = _POOL.submit(H.run_scenario, H.PRODUCT_LIB, name, workdir)


def job(workdir, kind, name):
    _start_all(workdir)
    return _JOBS[(kind, name)].result()


def scenario_trace(name, workdir):
    return job(workdir, "trace", name)


@pytest.fixture(scope="module")
def gold():
    return json.load(open(GOLD))


def _first_difference(got, want):
    for g, w in zip(got, want):
        if g == w:
            continue
        if g["stage"] != w["stage"]:
            return f"stage order: got '{g['stage']}', pinned build had '{w['stage']}'"
        if g["calls"] != w["calls"]:
            return f"stage '{g['stage']}': calls {g['calls']} vs validated {w['calls']}"
        for i, (a, b) in enumerate(zip(g["launches"], w["launches"])):
            if a != b:
                return f"stage '{g['stage']}', launch {i}: {a}  vs validated  {b}"
        return f"stage '{g['stage']}': same launches, but uploads / copies / allocations differ"
    return f"{len(got)} stages vs validated {len(want)}"


@pytest.mark.parametrize("scenario", H.SCENARIOS)
def test_host_side_matches_the_pinned_build(scenario, workdir, gold):
    lines = scenario_trace(scenario, workdir)
    assert H.check_pointers(lines) > 0
    got = H.digest(lines)
    want = gold["scenarios"][scenario]
    assert got == want, ("the library's host side differs from the pinned build (" + gold["library_sha256"][:12] + ", backed by " +
                         str(gold.get("validated_by")) + "): " + _first_difference(got, want) + " -- if intended, validate this build "
                         "(GPU suite, or tools/sim_suite.sh while no GPU is reachable), then tests/golden/make_launch_traces.py")


def test_launch_list_of_the_headline_workload(workdir):
    """c2 (cat_res64, B = 64): what one forward asks of the GPU, straight from the trace"""
    lines = scenario_trace("c2", workdir)
    st = dict(H.stages(lines))
    fwd = [H.parse_launch(ln) for ln in st["forward B=64"] if ln.startswith("launch ")]
    names = [d["name"] for d in fwd]
    assert len(fwd) == 95                                        # temb MLP + 94 layer launches
    assert names.count("conv_t32") == 34 and names.count("conv_s") == 51 and names.count("gn_small_kernel") == 2
    # the dominant kernel: 4-wave conv_t32 with 256-pixel tiles on the 64x64 layers = B * 16 tiles * 1 n-tile workgroups
    big = [d for d in fwd if d["name"] == "conv_t32" and d["g"] == "1024,1,1" and d["b"] == "256,1,1"]
    assert len(big) == 12
    # every launch of the forward is on the caller's stream (NULL here) and the sampling loop adds exactly one Euler step
    # per forward and one snapshot copy per masked step
    assert all(d["st"] == "(nil)" for d in fwd)
    loop = st["sample_iadb B=64 steps=3 snapshots at 1,2"]
    ln = [H.parse_launch(x) for x in loop if x.startswith("launch ")]
    # per step: 93 layer launches (the time embedding comes from the per-schedule table: one temb MLP + one projection GEMM for
    # the whole call) + the Euler update
    assert len(ln) == 3 * 94 + 2 and [d["name"] for d in ln].count("iadb_step_kernel") == 3
    assert sum(1 for x in loop if x.startswith("memcpy_async") and "src=0x" in x) == 2


def _same_toolchain(gold):
    from tests.golden.make_launch_traces import hipcc_version
    return gold.get("hipcc_version") in (None, hipcc_version())


def test_device_code_is_the_pinned_build(gold):
    """the gfx950 code objects inside the library are byte for byte those of the pinned build: together with the identical
    host trace above, a rebuilt library behaves exactly like the pinned one.  (Compiler output: skipped under another hipcc.)"""
    import hashlib
    from tests.hipmock.kernargs import code_objects
    if not _same_toolchain(gold):
        pytest.skip(f"the fixture was cut with '{gold.get('hipcc_version')}': another compiler's output is not comparable byte for byte")
    got = [hashlib.sha256(co).hexdigest() for co in code_objects(H.PRODUCT_LIB)]
    assert got == gold["device_code_sha256"], "device code differs from the pinned build: validate the new build (GPU suite / " \
                                              "tools/sim_suite.sh), then tests/golden/make_launch_traces.py"


def test_the_pinned_build_is_backed_by_a_committed_log(gold):
    """the fixture names what validated the pinned library -- a GPU-suite log or the simulator suite's log -- and that log,
    committed under profiles/, names the library's sha256 (the round-5 fixture pinned the build to itself)"""
    vb = gold.get("validated_by")
    assert vb, "tests/golden/launch_traces.json carries no 'validated_by'"
    path = vb.split(":", 1)[1] if vb.startswith("sim:") else vb
    full = os.path.join(ROOT, path)
    assert os.path.exists(full), f"{path} is not in the tree"
    assert gold["library_sha256"] in open(full).read(), f"{path} does not name the pinned library {gold['library_sha256'][:16]}"


def test_kernels_unchanged_since_the_driver_green_build(gold):
    """per KERNEL: which machine code is byte-identical to the last build the DRIVER ran green (round 3, commit f458ce0, library
    823a75b0...) is recorded when the fixture is cut; a later build may only shrink that list knowingly"""
    if not _same_toolchain(gold):
        pytest.skip("another hipcc: machine code is not comparable byte for byte")
    from tests.golden.make_launch_traces import kernel_hashes
    now = kernel_hashes(H.PRODUCT_LIB)
    assert now == gold["kernel_code_sha256"], sorted(k for k in now if gold["kernel_code_sha256"].get(k) != now[k])[:5]
    same, changed = gold["kernels_equal_to_r03_green"], gold["kernels_changed_since_r03_green"]
    assert gold["r03_green_library_sha256"].startswith("823a75b0")
    assert len(same) + len(changed) == len(now)
    # what changed since the driver-green build: the conv_t32 family (round 4: lean normalisation, pinned phases) and conv_in
    # (round 6: first-level width 64 swizzle fix) -- nothing else
    assert all("conv_t32" in k or "conv_in_kernel" in k for k in changed), [k for k in changed if "conv_t32" not in k and "conv_in" not in k]


def test_conv_t32_launches_compute_their_layers(workdir):
    """Every conv_t32 launch of a forward, evaluated on the CPU from its kernel arguments and uploaded tables (packed weights
    decoded, GroupNorm finalised from the partial sums, segments in order), equals the layer's definition on the ORIGINAL state-dict
    tensors: weight packing, K-step order, ln 2 fold, concat / upsample / shortcut segments, gamma / beta / bias wiring.  Host
    side of the dominant kernel only -- nothing here runs device code (tests/hipmock/check_conv_t32.py)."""
    out = job(workdir, "script", "check_conv_t32.py")
    assert "OK 34 conv_t32 launches" in out, out[-2000:]


def test_conv_s_launches_compute_their_layers(workdir):
    """The same for the <= 8x8 levels (tests/hipmock/check_conv_s.py): every conv_s launch evaluated from its step lists, round
    table, weight stream and epilogue requests equals the module's definition -- 3x3 / stride-2 / nearest-2x / 1x1 shortcut
    convolutions, q|k|v + softmax, to_out, and the GroupNorm(+SiLU) copies, whose consumers are found by value in the state
    dict and must have the requested group size."""
    out = job(workdir, "script", "check_conv_s.py")
    assert "OK 51 conv_s launches" in out, out[-2000:]


@pytest.mark.parametrize("case", list(REPLAY_CASES))
def test_replay_through_kernel_models_equals_the_oracle(case, workdir):
    """The engine's own launch list, executed on the CPU: every launch the library records for a forward (or two steps of an
    in-engine IADB / DDIM / conditional loop) is handed to a numpy model of its kernel's contract that reads and writes the
    very buffers the kernel would (tests/hipmock/exec_forward.py), and the result equals oracle/unet_oracle.py + the samplers'
    update rules on the same weights and inputs.  Covers the launch order and every buffer hand-over between launches (skip
    connections, concatenations, normalised copies, partial sums, split-K slabs, the per-schedule time-embedding table,
    snapshots) on top of the per-launch checks above, for BASELINE.json's layouts.  Host side only: no device code runs."""
    import numpy as np
    import torch
    from oracle import unet_oracle as UO
    from tests.hipmock.exec_forward import DA, DDIM, DG, T_IN
    cin, cout, res, B, mode = REPLAY_CASES[case]
    if mode == "vae":
        from oracle import vae_oracle as VO                  # AutoencoderKL decoder
    cfg, sd, out_dir, out = job(workdir, "replay", case)
    assert "OK replayed" in out, out[-2000:]
    load = lambda what: torch.from_numpy(np.load(os.path.join(out_dir, f"exec_{case}_{what}.npy")))
    x = load("x")
    if mode == "vae":
        want = VO.decode(sd, cfg, x)                         # (the C ABI takes latents already divided by the scaling factor)
    elif mode == "forward":
        want = UO.forward(sd, cfg, x, load("t"))
    elif mode in ("iadb", "cond"):
        extra = load("extra") if mode == "cond" else None
        snaps = []
        for s in range(2):                                   # utils.py:196-226 / iadb_bn.py:384-438 with explicit tables
            d = UO.forward(sd, cfg, x if extra is None else torch.cat([x, extra], 1), T_IN[s])
            x = x + DA[s] * d[:, :x.shape[1]]
            if cout == 2 * x.shape[1]:
                x = x + DG[s] * d[:, x.shape[1]:]
            snaps.append(x)
        want = torch.stack(snaps)
    else:
        for s in range(2):                                   # ddim_diffusers.py:674-681, eps-prediction, eta 0, clip 1
            t, sat, s1at, sap, s1ap = DDIM[5 * s:5 * s + 5]
            eps = UO.forward(sd, cfg, x, t)
            x0 = ((x - s1at * eps) / sat).clamp(-1.0, 1.0)
            x = sap * x0 + s1ap * eps
        want = x
    got = load("out")
    rel = float((got - want).double().norm() / want.double().norm())
    print(f"{case}: replay through the kernel models vs the oracle: rel-L2 {rel:.3e}")
    bar = 5e-3 if mode == "vae" else (1e-2 if case.endswith("bf16") else (1e-4 if case.endswith("f32") else 2e-3))       # the GPU tests' bars (test_gpu_vae.py, test_gpu_unet.py)
    assert rel <= bar, f"{case}: rel-L2 {rel:.3e}"


def _c2_forward_rel(workdir, tag):
    import numpy as np
    import torch
    from oracle import unet_oracle as UO
    cfg, sd, out_dir, out = job(workdir, "replay", tag)
    x = torch.from_numpy(np.load(os.path.join(out_dir, "exec_c2_x.npy")))
    t = torch.from_numpy(np.load(os.path.join(out_dir, "exec_c2_t.npy")))
    want = UO.forward(sd, cfg, x, t)
    got = torch.from_numpy(np.load(os.path.join(out_dir, "exec_c2_out.npy")))
    return out, float((got - want).double().norm() / want.double().norm())


@pytest.mark.parametrize("switch", ["BNDM_NO_TAIL", "BNDM_NO_FUSED"])
def test_fallback_paths_replay_to_the_oracle(switch, workdir):
    """The kept fallbacks (<= 8x8 levels on implicit GEMM + gn_small; no conv_t32 at all) build other launch
    lists from the same graph -- replayed the same way (c2 layout, batch 2)"""
    out, rel = _c2_forward_rel(workdir, switch)
    assert ("conv_s" if switch == "BNDM_NO_TAIL" else "conv_t32") not in out, out[-400:]
    assert rel <= 2e-3, f"{switch}: rel-L2 {rel:.3e}"


def test_the_benchmarked_kernel_set_replays_to_the_oracle(workdir):
    """The handle bench.py builds (max_batch 64: conv_t32<TH=16>, conv_s<TM=128> -- tile variants follow the handle's batch)
    called at batch 2: the benchmarked launch list, replayed"""
    out, rel = _c2_forward_rel(workdir, "mb64")
    assert "conv_t32<TH=16>" in out and "conv_s<TM=128>" in out, out[-400:]
    assert rel <= 2e-3, f"rel-L2 {rel:.3e}"
