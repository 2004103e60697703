"""Per-kernel hash of the gfx950 machine code inside one or two builds of the library: which kernels did a change touch?
    python tools/kernel_isa_hash.py libA.so [libB.so]      (with two libraries: prints only the kernels that differ / are new)
    python tools/kernel_isa_hash.py --dump <substring> lib.so > kernel.s      (disassembly of the kernels whose name matches)"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.hipmock.kernargs import code_objects  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def kernels(lib):
    """{symbol: [instruction text lines]} over every code object of the library"""
    out = {}
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".o") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--no-leading-addr", f.name], check=True, capture_output=True, text=True).stdout
        cur = None
        for ln in txt.splitlines():
            m = re.match(r"^[0-9a-f]* ?<([^>]+)>:$", ln.strip())
            if m:
                cur = m.group(1)
                out[cur] = []
            elif cur and ln.strip():
                out[cur].append(re.sub(r"\s*//.*$", "", ln.strip()))
    return out


def main():
    if sys.argv[1] == "--dump":
        for k, v in kernels(sys.argv[3]).items():
            if sys.argv[2] in k:
                print(f"<{k}>:")
                print("\n".join(v))
        return
    ks = [kernels(p) for p in sys.argv[1:]]
    h = [{k: hashlib.sha256("\n".join(v).encode()).hexdigest()[:12] for k, v in d.items()} for d in ks]
    if len(h) == 1:
        for k in sorted(h[0]):
            print(h[0][k], len(ks[0][k]), k)
        return
    same = 0
    for k in sorted(set(h[0]) | set(h[1])):
        a, b = h[0].get(k), h[1].get(k)
        if a == b:
            same += 1
        else:
            print(f"{a or 'absent':12s} {b or 'absent':12s} {len(ks[0].get(k, []))} -> {len(ks[1].get(k, []))} instr  {k}")
    print(f"{same} kernels identical")


if __name__ == "__main__":
    main()
