"""CPU test that belongs to tools/experiments/lanes.patch (apply the patch first; not collected by `pytest`, see pytest.ini)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def test_lanes_options_reach_the_model_and_the_cli():
    """lanes are host-side plumbing down to bndm_unet_set_lanes (include/bndm_hip.h); no GPU needed to build the module"""
    import pytest
    sys.path.insert(0, ROOT)
    from bndm_amd.sampler import get_model
    from bndm_amd import _lib, cli_iadb
    if "bndm_unet_set_lanes" not in _lib.SIGNATURES:
        # the product has no lanes (its UNet2DModel rejects lane* arguments so that nothing passes vacuously): this test belongs to a tree
        # with tools/experiments/lanes.patch applied
        pytest.skip("tools/experiments/lanes.patch is not applied to this tree")
    m = get_model(3, 6, 64, lanes=4, lane_cus=True, lane_threads=True, lane_stagger=False)
    assert (m.lanes, m.lane_cus, m.lane_threads, m.lane_stagger) == (4, True, True, False)
    assert get_model(3, 6, 64).lanes == 1                                   # default: one chain
    assert _lib.SIGNATURES["bndm_unet_set_lanes"][1] == [_lib._vp, _lib._i, _lib._i]
    src = open(os.path.join(ROOT, "bndm_amd", "cli_iadb.py")).read()
    assert "--lanes" in src and "--lane_cus" in src


# ---- host side of a library built WITH the patch (tools/build_candidates.sh -> tools/lib_lanes.so) under the recording HIP
# stand-in of tests/hipmock: no GPU needed
import json

import pytest

sys.path.insert(0, ROOT)
from tests.hipmock import harness as H  # noqa: E402

LANES_LIB = os.path.join(ROOT, "tools", "lib_lanes.so")
needs_lib = pytest.mark.skipif(not os.path.exists(LANES_LIB), reason="tools/lib_lanes.so not built (tools/build_candidates.sh)")


@needs_lib
@pytest.mark.parametrize("scenario", ["c2", "c3", "c5", "cond"])
def test_one_lane_is_the_product_host_side(scenario, tmp_path):
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "launch_traces.json")))
    lines = H.run_scenario(LANES_LIB, scenario, str(tmp_path))
    assert H.digest(lines) == gold["scenarios"][scenario]


@needs_lib
@pytest.mark.parametrize("lanes,flags", [(2, 0), (4, 0), (4, 4), (2, 2), (4, 1)])
def test_chains_are_copies_of_each_other_on_their_own_buffers(lanes, flags, tmp_path):
    lines = H.run_scenario(LANES_LIB, "c2", str(tmp_path), lanes, flags)
    assert H.check_pointers(lines) > 0                          # every lane's pointers inside the (lanes x) allocations
    st = dict(H.stages(lines))
    loop = [H.parse_launch(x) for x in st["sample_iadb B=64 steps=3 snapshots at 1,2"] if x.startswith("launch ")]
    by_stream = {}
    for d in loop:
        by_stream.setdefault(d["st"], []).append(d)
    once = [d for d in loop if d["name"] == "temb_mlp_kernel"]
    assert len(once) == 1                                       # the per-schedule table stays single
    chains = []
    for s, ds in by_stream.items():
        ds = [d for d in ds if d["name"] != "temb_mlp_kernel" and not (d["name"] == "conv_igemm" and d is ds[1] and s == once[0]["st"])]
        if ds:                                                      # (CU shares: the caller's stream only carries the table)
            chains.append([(d["name"], d["g"], d["b"], d["lds"]) for d in ds])
    assert len(chains) == lanes
    for c in chains[1:]:
        assert c == chains[0]                                   # same kernels, grids and LDS sizes in every chain
    # no two chains share a written buffer: the output pointer of every conv_t32 launch (FusedArgs::out) differs between chains
    assert len({d["args"][0] for d in loop if d["name"] == "iadb_step_kernel"}) == lanes
