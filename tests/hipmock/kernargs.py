"""TEST INFRASTRUCTURE: the explicit kernel-argument sizes of every kernel in a libbndm_hip.so, read from the AMDGPU metadata
notes of the gfx950 code objects inside its .hip_fatbin section (clang offload bundles), for tests/hipmock/hipmock.cpp.

    python tests/hipmock/kernargs.py <lib.so> <out.txt>      one line per kernel: <symbol> <nargs> <size>...
"""
import os
import re
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



# ---- which bytes of its kernel arguments does a kernel READ? ---------------------------------------------------------------------------
# Struct arguments are passed by value: their padding bytes (and fields a kernel never looks at) are whatever the host left there.  The
# trace digests (harness.py) compare argument BYTES, so those must not take part.  The code object's metadata has no field layout, but the
# machine code says which kernarg bytes are fetched: every `s_load_dword*` through the kernarg pointer s[0:1], or through a pointer derived
# from it by `s_add_u32 sA, s0, imm ; s_addc_u32 sB, s1, 0 | -1`.  A kernel that does anything else with the pointer (vector access, a
# register displacement, pointer arithmetic in a loop) is left undecided: None, all of its bytes are compared.
_NDW = {"s_load_dword": 1, "s_load_dwordx2": 2, "s_load_dwordx4": 4, "s_load_dwordx8": 8, "s_load_dwordx16": 16}
_WRITES_SGPR = ("s_", "v_readfirstlane", "v_readlane", "v_cmp")
_WRITES_SGPR_2ND = ("v_mad_u64", "v_mad_i64", "v_add_co", "v_addc_co", "v_sub_co", "v_subb", "v_subrev_co", "v_subbrev", "v_div_scale")


def _sregs(tok):
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    return [int(m.group(1))] if m else []


def read_mask(k):
    """k: tests.gfx950sim.loader.KernelInfo -> bytes (1 = read) over the kernarg segment, or None"""
    mask = bytearray(k.kernarg_size + 64)
    bases = {(0, 1): 0}               # SGPR pair -> its constant displacement from the kernarg pointer
    half = {}                         # "sA" -> displacement, between the s_add_u32 and its s_addc_u32

    def forget(regs):
        for r in regs:
            for p in [p for p in bases if r in p]:
                del bases[p]
            for a in [a for a in half if int(a[1:]) == r]:
                del half[a]

    for ins in k.insts:
        m, ops = ins.mnem, ins.ops
        if m in _NDW and len(ops) >= 3:
            br = tuple(_sregs(ops[1]))
            # (a load through s[0:1] counts wherever it stands, even behind an overwrite of s0 / s1 in address order: blocks need not be
            # laid out in execution order, and more bytes "read" only means fewer bytes masked)
            if br in bases or br == (0, 1):
                try:
                    off = int(ops[2], 0)
                except ValueError:
                    return None
                lo = bases.get(br, 0) + off
                for i in range(max(lo, 0), min(lo + 4 * _NDW[m], len(mask))):
                    mask[i] = 1
            forget(_sregs(ops[0]))
            continue
        if m == "s_add_u32" and ops[1] == "s0" and (0, 1) in bases:
            try:
                v = int(ops[2], 0)
            except ValueError:
                return None
            if _sregs(ops[0])[0] in (0, 1):
                return None
            half[ops[0]] = v - (1 << 32) if v >= 1 << 31 else v
            continue
        if m == "s_addc_u32" and ops[1] == "s1" and ops[2] in ("0", "-1") and (0, 1) in bases:
            d = _sregs(ops[0])[0]
            a = [r for r in half if int(r[1:]) + 1 == d]
            if not a:
                return None
            bases[(d - 1, d)] = half.pop(a[0])
            continue
        tracked = {r for p in bases for r in p} | {int(a[1:]) for a in half}
        used = {r for t in ops[1:] for tok in re.findall(r"s\[\d+:\d+\]|\bs\d+\b", t) for r in _sregs(tok)}
        if used & tracked:
            return None                # the pointer goes somewhere this scan does not follow
        if ops and m.startswith(_WRITES_SGPR):
            forget(_sregs(ops[0]))
        if len(ops) > 1 and m.startswith(_WRITES_SGPR_2ND):
            forget(_sregs(ops[1]))
    return bytes(mask[:k.kernarg_size])


def read_masks(lib):
    """{kernel symbol: [bytes-read mask of each explicit argument (bytes of 0 / 1)], or None when undecided}"""
    from tests.gfx950sim import loader
    out = {}
    for name, k in loader.load_library(lib).items():
        mk = read_mask(k)
        expl = [a for a in k.args if not str(a[".value_kind"]).startswith("hidden_")]
        out[name] = None if mk is None else [bytes(mk[int(a[".offset"]):int(a[".offset"]) + int(a[".size"])]) for a in expl]
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
