"""CPU checks of candidate libraries' HOST side under the recording HIP runtime (tests/hipmock); not collected by `pytest`
(pytest.ini), run by name:  python -m pytest tools/experiments/extra_tests/test_candidate_hosts.py -q"""
import json
import os
import struct
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from tests.hipmock import harness as H  # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "launch_traces.json")))


def lib(name):
    p = os.path.join(ROOT, "tools", name)
    if not os.path.exists(p):
        pytest.skip(f"tools/{name} not built (tools/build_candidates.sh)")
    return p


@pytest.mark.parametrize("name", ["lib_v9.so", "lib_v12.so", "lib_v13.so"])
@pytest.mark.parametrize("scenario", ["c2", "c4"])
def test_kernel_only_candidates_keep_the_host_side(name, scenario, tmp_path):
    """staged 1x1 chunks / scalar chunk descriptors / first-round write-back change device code only: same launches, same
    argument bytes, same uploads as the product (LDS sizes may differ: they are part of the kernel)"""
    got = H.digest(H.run_scenario(lib(name), scenario, str(tmp_path)))
    want = GOLD["scenarios"][scenario]
    strip = lambda recs: [[" ".join(x.split(" ")[:3]) for x in r["launches"]] for r in recs]      # name, grid, block
    assert strip(got) == strip(want)
    assert [r["calls"] for r in got] == [r["calls"] for r in want]


@pytest.mark.parametrize("scenario,cout", [("c2", 6), ("c5", 8), ("cond", 3)])
def test_euler_step_in_the_head_launch(scenario, cout, tmp_path):
    """lib_v17 (head_conv_kernel.patch): no iadb_step launch in the in-engine loop; the head_conv launch carries the sampler state and the
    step's da / dg (HeadArgs offsets: Cout 88, ex 96, eda 104, edg 108, eC 112); forwards outside the loop store d (ex == 0)"""
    lines = H.run_scenario(lib("lib_v17.so"), scenario, str(tmp_path))
    H.check_pointers(lines)
    for mark, body in H.stages(lines):
        ls = [H.parse_launch(x) for x in body if x.startswith("launch ")]
        heads = [d for d in ls if d["name"] == "head_conv"]
        if mark.startswith("sample_iadb"):
            x = int([ln for ln in body if ln.startswith("malloc")][0].split()[1], 16)
            assert not [d for d in ls if d["name"] == "iadb_step_kernel"]
            steps = 3 if "steps=3" in mark else 2
            assert len(heads) == steps
            for s, d in enumerate(heads):
                a = d["args"][0]
                assert len(a) == 120
                co, = struct.unpack_from("<i", a, 88)
                ex, = struct.unpack_from("<Q", a, 96)
                da, dg = struct.unpack_from("<ff", a, 104)
                ec, = struct.unpack_from("<i", a, 112)
                assert ex == x and co == cout and ec == (cout if scenario == "cond" else cout // 2)
                assert da == struct.unpack("<f", struct.pack("<f", -1.0 / steps))[0] and dg == struct.unpack("<f", struct.pack("<f", -0.5 / steps))[0]
        elif mark.startswith("forward"):
            assert len(heads) == 1 and struct.unpack_from("<Q", heads[0]["args"][0], 96)[0] == 0


# ---- whole loops replayed on the CPU through a candidate's own launch list (tests/hipmock/exec_forward.py) vs the oracle ----
def _replay(libname, case, tmp_path, env=None):
    import numpy as np
    import torch
    from oracle import unet_oracle as UO
    from tests.hipmock.exec_forward import DA, DG, T_IN
    from tests.test_launch_trace import REPLAY_CASES
    cin, cout, res, B, mode = REPLAY_CASES[case]
    assert mode == "iadb"
    cfg = UO.make_config(res, cin, cout)
    sd = UO.init_params(cfg, seed=0, perturb_norm=0.1)
    wd = str(tmp_path)
    wfile = os.path.join(wd, "w.npz")
    np.savez(wfile, **{k: v.numpy() for k, v in sd.items()})
    old = dict(os.environ)
    os.environ.update(env or {})
    try:
        out = H.run_script("exec_forward.py", lib(libname), wd, wd, case, wfile)
    finally:
        os.environ.clear()
        os.environ.update(old)
    os.remove(wfile)
    assert "OK replayed" in out
    x = torch.from_numpy(np.load(os.path.join(wd, f"exec_{case}_x.npy")))
    snaps = []
    for s in range(2):
        d = UO.forward(sd, cfg, x, T_IN[s])
        x = x + DA[s] * d[:, :3] + DG[s] * d[:, 3:]
        snaps.append(x)
    got = torch.from_numpy(np.load(os.path.join(wd, f"exec_{case}_out.npy")))
    rel = float((got - torch.stack(snaps)).double().norm() / torch.stack(snaps).double().norm())
    assert rel <= 2e-3, rel
    return out


def test_euler_step_candidate_loop_equals_the_oracle(tmp_path):
    """lib_v17: the head_conv launch's contract (x updated in place, no iadb_step launch) replayed end to end through the numpy models"""
    out = _replay("lib_v17.so", "c2loop", tmp_path)
    base = _replay("../bndm_amd/libbndm_hip.so", "c2loop", tmp_path)
    count = lambda o: int(o.split("replayed")[1].split()[0])
    assert count(out) == count(base) - 2                     # one launch fewer per step


@pytest.mark.parametrize("lanes,flags,batch", [(2, 0, 2), (2, 4, 2), (2, 2, 2), (4, 0, 4), (3, 2, 3)])
def test_lanes_candidate_loop_equals_the_oracle(lanes, flags, batch, tmp_path):
    """lib_lanes: chains on their own (smaller) buffer copies, replayed in enqueue order, give the one-chain result"""
    _replay("lib_lanes.so", "c2loop", tmp_path, {"EXEC_LANES": str(lanes), "EXEC_LANE_FLAGS": str(flags), "EXEC_BATCH": str(batch)})


@pytest.mark.parametrize("case", ["c2", "c4"])
def test_pair_statistics_candidate_forward_equals_the_oracle(case, tmp_path):
    """lib_v8 (round-4 patch): GroupNorm partial sums per channel pair in every producer and consumer -- the forward replayed with
    the models switched to that layout (EXEC_PAIRSTATS=1) reproduces the oracle; with the product's layout it must NOT (the
    switch really is the layout)"""
    import numpy as np
    import torch
    from oracle import unet_oracle as UO
    from tests.test_launch_trace import REPLAY_CASES
    cin, cout, res, B, mode = REPLAY_CASES[case]
    cfg = UO.make_config(res, cin, cout)
    sd = UO.init_params(cfg, seed=0, perturb_norm=0.1)
    wd = str(tmp_path)
    wfile = os.path.join(wd, "w.npz")
    np.savez(wfile, **{k: v.numpy() for k, v in sd.items()})
    rels = {}
    for pair in ("1", "0", "th32"):
        os.environ["EXEC_PAIRSTATS"] = "0" if pair == "0" else "1"
        if pair == "th32":                                   # 512-pixel tiles wherever the shape allows: other slab counts
            os.environ["BNDM_TH32_MIN"] = "1"
            os.environ["BNDM_TH16_MIN"] = "1"                # (256-pixel tiles first, also at this small batch)
        try:
            out = H.run_script("exec_forward.py", lib("lib_v8.so"), wd, wd, case, wfile)
        except AssertionError:
            rels[pair] = float("inf")                        # (the wrong layout may also run out of a buffer's bounds)
            continue
        finally:
            os.environ.pop("EXEC_PAIRSTATS", None)
            os.environ.pop("BNDM_TH32_MIN", None)
            os.environ.pop("BNDM_TH16_MIN", None)
        if pair == "th32":
            assert "TH=32" in out, "BNDM_TH32_MIN=1 selected no conv_t32<TH=32> launch"
        x = torch.from_numpy(np.load(os.path.join(wd, f"exec_{case}_x.npy")))
        t = torch.from_numpy(np.load(os.path.join(wd, f"exec_{case}_t.npy")))
        want = UO.forward(sd, cfg, x, t)
        got = torch.from_numpy(np.load(os.path.join(wd, f"exec_{case}_out.npy")))
        rels[pair] = float((got - want).double().norm() / want.double().norm())
    os.remove(wfile)
    assert rels["1"] <= 2e-3 and rels["th32"] <= 2e-3 and rels["0"] > 1e-2, rels
