"""Regenerates tests/golden/launch_traces.json: digests of the host-side traces (tests/hipmock) of the library's C ABI for
BASELINE.json's configurations, the sha256 of its gfx950 code objects and of every kernel's machine code.  A fixture is data:
kernel names, grids, argument hashes -- no source text.

    python tests/golden/make_launch_traces.py [lib.so] [--green r03_lib.so] [--validated-by profiles/rNN_gpu_tests.log | sim:profiles/rNN_sim_suite.log]

What the fixture is: the PINNED build -- every later build's host side (launch list, grids, kernel arguments, uploads, buffer
layout) and device code are compared with it (tests/test_launch_trace.py).  What backs the pinned build is recorded in the
fixture itself ("validated_by"): a committed GPU-suite log that contains this library's sha256, or -- while no GPU is
reachable -- the simulator suite's log (tests/gfx950sim: the machine code executed on the CPU against oracle/).
"kernels_equal_to_r03_green" lists the kernels whose machine code is byte-identical to the last DRIVER-green build's
(commit f458ce0, library sha 823a75b0...; rebuild it with `git archive f458ce0 | tar -x -C /tmp/r03 && make -C /tmp/r03/bndm_amd/csrc`)."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.hipmock import harness as H  # noqa: E402


def kernel_hashes(lib):
    """{kernel symbol: sha256 of its instruction words} (position-independent: encodings, not addresses)"""
    from tests.gfx950sim import loader
    out = {}
    for name, k in loader.load_library(lib).items():
        h = hashlib.sha256()
        for ins in k.insts:
            for wd in ins.enc:
                h.update(wd.to_bytes(4, "little"))
        out[name] = h.hexdigest()
    return out


def hipcc_version():
    try:
        txt = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout
        return next((ln.strip() for ln in txt.splitlines() if "HIP version" in ln), txt.splitlines()[0].strip())
    except Exception:                                        # noqa: BLE001
        return "unknown"


def main():
    args = sys.argv[1:]
    green = validated = None
    if "--green" in args:
        i = args.index("--green")
        green = args[i + 1]
        del args[i:i + 2]
    if "--validated-by" in args:
        i = args.index("--validated-by")
        validated = args[i + 1]
        del args[i:i + 2]
    lib = os.path.abspath(args[0]) if args else H.PRODUCT_LIB
    from tests.hipmock.kernargs import code_objects
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()
    kh = kernel_hashes(lib)
    out = {"library_sha256": sha, "hipcc_version": hipcc_version(), "validated_by": validated,
           # one hash per .hip source: the gfx950 code objects of the pinned build
           "device_code_sha256": [hashlib.sha256(co).hexdigest() for co in code_objects(lib)],
           "kernel_code_sha256": kh, "scenarios": {}}
    if validated:
        path = validated.split(":", 1)[1] if validated.startswith("sim:") else validated
        assert sha in open(os.path.join(ROOT, path)).read(), f"{path} does not mention the library's sha256 {sha}"
    if green:
        gh = kernel_hashes(os.path.abspath(green))
        out["r03_green_library_sha256"] = hashlib.sha256(open(green, "rb").read()).hexdigest()
        out["kernels_equal_to_r03_green"] = sorted(k for k, v in kh.items() if gh.get(k) == v)
        out["kernels_changed_since_r03_green"] = sorted(k for k, v in kh.items() if gh.get(k) != v)
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
