"""TEST INFRASTRUCTURE: the explicit kernel-argument sizes of every kernel in a libbndm_hip.so, read from the AMDGPU metadata
notes of the gfx950 code objects inside its .hip_fatbin section (clang offload bundles), for tests/hipmock/hipmock.cpp.

    python tests/hipmock/kernargs.py <lib.so> <out.txt>      one line per kernel: <symbol> <nargs> <size>...
"""
import os
import struct
import subprocess
import sys
import tempfile

import yaml

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        blob = open(fat, "rb").read()
    pos = blob.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, q)
            ident = blob[q + 24:q + 24 + idlen].decode()
            q += 24 + idlen
            if "gfx950" in ident and size:
                yield blob[pos + off:pos + off + size]
        pos = blob.find(MAGIC, pos + len(MAGIC))


def kernel_args(lib):
    out = {}
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".o") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], check=True, capture_output=True, text=True).stdout
        if "amdhsa.kernels" not in txt:
            continue
        doc = txt[txt.index("amdhsa.kernels"):]
        doc = doc.split("\n...")[0]
        meta = yaml.safe_load(doc)
        for k in meta["amdhsa.kernels"]:
            args = [a for a in k.get(".args", []) if not str(a[".value_kind"]).startswith("hidden_")]
            out[k[".name"]] = [(int(a[".offset"]), int(a[".size"])) for a in args]
    return out


def main():
    lib, dst = sys.argv[1], sys.argv[2]
    ka = kernel_args(lib)
    with open(dst, "w") as f:
        for name, args in sorted(ka.items()):
            f.write(f"{name} {len(args)} " + " ".join(str(s) for _, s in args) + "\n")
    print(f"{len(ka)} kernels")


if __name__ == "__main__":
    main()
