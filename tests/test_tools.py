"""CPU tests of the measurement / desk-check tools that carry conclusions in DESIGN.md (no GPU, no library calls)."""
import csv
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, path))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_staged_shortcut_index_arithmetic():
    """tools/experiments/conv_t32_shortcut_stages.patch: DMA placement, fragment reads and banks of the staged 1x1 loop"""
    m = _load("tools/experiments/check_shortcut_stages.py", "check_shortcut_stages")
    for th in (8, 16):
        for nw in (4, 8):
            m.check(th, nw)


def test_overlap_tool_on_a_synthetic_trace(tmp_path):
    d = tmp_path / "x"
    d.mkdir()
    with open(d / "1_kernel_trace.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id"])
        w.writerow(["conv_t32<a>", 0, 100000, "1"])
        w.writerow(["conv_t32<a>", 20000, 120000, "2"])        # 80 % of each beside the other queue
        w.writerow(["conv_s<x>", 130000, 140000, "1"])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "overlap.py"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "3 kernels over 0.140 ms" in out
    lines = {ln.split()[0]: ln.split() for ln in out.splitlines() if ln.startswith(("conv_t32", "conv_s"))}
    assert lines["conv_t32"][1] == "0" and lines["conv_t32"][3] == "2"          # both counted as running beside another queue
    assert lines["conv_s"][1] == "1"
    two = [ln for ln in out.splitlines() if ln.startswith("distinct queues")][0]
    assert "2:  57.1%" in two                                                    # 80 us of 140


def test_no_bit_cast_of_a_vector_element_in_the_kernels():
    """With this toolchain (clang 22 of ROCm 7.2) `__builtin_bit_cast(T, vec[e])` on an ext-vector ELEMENT reads element 0 -- four
    ds_bpermute of different data fold into one (found on the instruction-level simulator in round 6, reduced case in
    profiles/r06_README.md).  Kernel sources must go through a scalar temporary."""
    import glob
    import re
    pat = re.compile(r"__builtin_bit_cast\(\s*[\w ]+,\s*\w+\s*(\[[^\]]+\]\s*)+\)")
    hits = []
    for f in glob.glob(os.path.join(ROOT, "bndm_amd", "csrc", "*.h*")):
        for i, ln in enumerate(open(f), 1):
            if pat.search(ln.split("//")[0]):
                hits.append(f"{os.path.basename(f)}:{i}: {ln.strip()}")
    assert not hits, "bit_cast of a subscripted value (vector element?): use a scalar temporary\n" + "\n".join(hits)
