"""Profiling aid: run bench.py against ANOTHER build of the library (same-box A/B of a kernel change):
    python tools/bench_with_lib.py tools/libbndm_old.so [bench.py flags...]
The product loader has no library override; this script points it at the given file explicitly."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bndm_amd import _lib   # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import bench                # noqa: E402

sys.argv = ["bench.py"] + sys.argv[2:]
bench.main()
