"""Profiling aid used by tools/ablate.sh: bench.py --profile-only against the ABLATION build of the library
(tools/libbndm_ablate.so, compiled with -DBNDM_ABLATION).  The product loader has no library override; this script
points it at the profiling build explicitly."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bndm_amd import _lib   # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "libbndm_ablate.so")
import bench                # noqa: E402

sys.argv = ["bench.py", "--profile-only", "--no-cpu-baseline", "--allow-ablation"] + sys.argv[1:]
bench.main()
