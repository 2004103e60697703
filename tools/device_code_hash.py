"""sha256 of every gfx950 code object inside a libbndm_hip.so (one per .hip source).  Two builds with the same hashes run the same
device code; together with an identical host trace (tests/test_launch_trace.py) that is the whole behaviour of the library.
    python tools/device_code_hash.py libA.so [libB.so ...]"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.hipmock.kernargs import code_objects  # noqa: E402

for lib in sys.argv[1:]:
    hs = [hashlib.sha256(co).hexdigest()[:16] for co in code_objects(lib)]
    print(f"{lib}: {' '.join(hs)}   all={hashlib.sha256(' '.join(hs).encode()).hexdigest()[:16]}")
