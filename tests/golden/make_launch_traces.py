"""Regenerates tests/golden/launch_traces.json: digests of the host-side traces (tests/hipmock) of the library's C ABI for
BASELINE.json's configurations.  A fixture is data: kernel names, grids, argument hashes -- no source text.

    python tests/golden/make_launch_traces.py [lib.so]

Generate it ONLY from a library whose full `-m gpu` suite has passed on an MI355X: the fixture then pins every later build's
host side (launch list, grids, kernel arguments, uploads, buffer layout) to the one that ran.  The committed file was written
from the build of commit 996594f's kernel sources (sha256 in the file), the last one the full suite ran on (round 4)."""
import hashlib
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.hipmock import harness as H  # noqa: E402


def main():
    lib = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else H.PRODUCT_LIB
    from tests.hipmock.kernargs import code_objects
    out = {"library_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),
           # one hash per .hip source: the gfx950 code objects the validated build ran
           "device_code_sha256": [hashlib.sha256(co).hexdigest() for co in code_objects(lib)], "scenarios": {}}
    with tempfile.TemporaryDirectory() as td:
        for s in H.SCENARIOS:
            lines = H.run_scenario(lib, s, td)
            H.check_pointers(lines)
            out["scenarios"][s] = H.digest(lines)
            print(s, sum(len(st["launches"]) for st in out["scenarios"][s]), "launches")
    with open(os.path.join(ROOT, "tests", "golden", "launch_traces.json"), "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))


if __name__ == "__main__":
    main()
