"""TEST INFRASTRUCTURE: sensitivity of the simulator -- the product library with counted waits of one kernel loosened in the
parsed instruction stream (the binary is not touched) must fail: hazards (reads of LDS bytes / registers still in flight) or a
wrong result.  Shows that a kernel whose `s_waitcnt vmcnt(N)` is off by a few is caught deterministically."""
import os
import re

import numpy as np


def run_with_loosened_waits(workdir, kernel_substr, add=3, case="lat256", batch=1):
    import torch
    from tests import test_launch_trace as T
    from tests.gfx950sim import suite
    from tests.hipmock import harness as H
    torch.set_num_threads(4)
    cfg, key, make = T._case_network(case)
    sd, wfile = T.oracle_weights(workdir, key, make)
    out_dir = os.path.join(workdir, "loosened")
    env = dict(EXEC_SIM="1", EXEC_BATCH=str(batch), GFX950SIM_PROCS="8", GFX950SIM_LOOSEN=f"{kernel_substr}:{add}", OMP_NUM_THREADS="4",
               GFX950SIM_STRICT="1")                  # the first hazard ends the run (what is asked is whether there is one)
    try:
        out = H.run_script("exec_forward.py", H.PRODUCT_LIB, out_dir, out_dir, case, wfile, env=env, mockdir=workdir)
    except AssertionError as e:                      # the run died (an out-of-bounds access, a runaway loop): also a catch
        return dict(hazards=1, rel=float("nan"), detail=str(e)[-1500:])
    m = re.search(r"hazards (\d+)", out)
    got, want = suite.expected(case, sd, cfg, out_dir)
    rel = float((got - want).double().norm() / want.double().norm())
    assert "loosened" in out and int(re.search(r"loosened (\d+) counted", out).group(1)) > 0, out[-1500:]
    return dict(hazards=int(m.group(1)) if m else -1, rel=rel if np.isfinite(rel) else float("nan"), detail=out[-800:])
