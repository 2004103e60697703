"""TEST INFRASTRUCTURE: reads the gfx950 code objects out of a libbndm_hip.so -- disassembly (llvm-objdump), kernel
descriptors (the .kd symbols of each ELF) and the AMDGPU metadata notes -- for the instruction-level simulator in sim.py.
Nothing here is product code; the product never imports it."""
import hashlib
import os
import pickle
import re
import struct
import subprocess
import sys
import tempfile

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.hipmock.kernargs import code_objects  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin"
CACHE = os.environ.get("GFX950SIM_CACHE", os.path.join(tempfile.gettempdir(), "gfx950sim_cache"))


class Inst:
    __slots__ = ("addr", "size", "mnem", "ops", "mods", "text", "fn", "base", "enc")

    def __init__(self, addr, size, mnem, ops, mods, text):
        self.addr, self.size, self.mnem, self.ops, self.mods, self.text = addr, size, mnem, ops, mods, text
        self.fn = None
        self.base = None
        self.enc = None

    def __repr__(self):
        return f"{self.addr:#x}: {self.text}"


class KernelInfo:
    """one kernel: instruction list + launch-relevant descriptor fields"""

    def __init__(self, name):
        self.name = name
        self.insts = []
        self.index = {}                # address -> index into insts
        self.args = []                 # metadata .args (dicts)
        self.kernarg_size = 0
        self.lds_static = 0
        self.scratch = 0
        self.rsrc1 = self.rsrc2 = self.rsrc3 = 0
        self.code_props = 0
        self.preload = 0
        self.entry = 0
        self.vgpr_count = self.agpr_count = self.sgpr_count = 0


def _split_top(s, sep):
    """split at `sep` outside [...] and (...)"""
    out, depth, cur = [], 0, []
    for ch in s:
        if ch in "[(":
            depth += 1
        elif ch in "])":
            depth -= 1
        if ch == sep and depth == 0:
            out.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    out.append("".join(cur))
    return out


_LINE = re.compile(r"^\t(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*((?:[0-9A-Fa-f]{8}\s*)+)")


def parse_line(line):
    m = _LINE.match(line)
    if not m:
        return None
    mnem, rest, addr, words = m.group(1), m.group(2), int(m.group(3), 16), m.group(4).split()
    words = [w for w in words if re.fullmatch(r"[0-9A-Fa-f]{8}", w)]
    ops, mods = [], []
    if mnem == "s_waitcnt":
        mods = rest.split()
    elif rest:
        parts = [p.strip() for p in _split_top(rest, ",")]
        last = [t for t in _split_top(parts[-1], " ") if t]
        parts[-1] = last[0] if last else ""
        mods = last[1:]
        # a modifier-only tail ("s_setprio 1" has none; "buffer_... 0 offen lds" handled above)
        ops = [p for p in parts if p != ""]
    ins = Inst(addr, 4 * len(words), mnem, ops, mods, (mnem + " " + rest).strip())
    ins.enc = [int(w, 16) for w in words]
    return ins


def _elf_kd(co):
    """-> {kernel name: 64 descriptor bytes} from an ELF64 code object (symbols '<name>.kd')"""
    (shoff,) = struct.unpack_from("<Q", co, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", co, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", co, shoff + i * shentsize)
        secs.append(dict(name=name, type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
    out = {}
    for s in secs:
        if s["type"] not in (2, 11):       # SHT_SYMTAB, SHT_DYNSYM
            continue
        strs = secs[s["link"]]
        for k in range(s["size"] // 24):
            st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", co, s["off"] + 24 * k)
            end = co.index(b"\0", strs["off"] + st_name)
            nm = co[strs["off"] + st_name:end].decode()
            if nm.endswith(".kd") and 0 < st_shndx < len(secs):
                sec = secs[st_shndx]
                o = sec["off"] + (st_value - sec["addr"])
                out[nm[:-3]] = (co[o:o + 64], st_value)
    return out


def _metadata(path):
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], check=True, capture_output=True, text=True).stdout
    if "amdhsa.kernels" not in txt:
        return []
    doc = txt[txt.index("amdhsa.kernels"):].split("\n...")[0]
    return yaml.safe_load(doc)["amdhsa.kernels"]


def parse_code_object(co):
    """one gfx950 ELF code object (bytes) -> {kernel symbol: KernelInfo}"""
    kernels = {}
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(co)
        f.flush()
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", f.name], check=True, capture_output=True,
                             text=True).stdout
        meta = _metadata(f.name)
    kds = _elf_kd(co)
    cur = None
    for line in dis.splitlines():
        m = re.match(r"^([0-9a-f]{16}) <(.+)>:$", line)
        if m:
            cur = KernelInfo(m.group(2))
            cur.entry = int(m.group(1), 16)
            kernels[cur.name] = cur
            continue
        if cur is None:
            continue
        ins = parse_line(line)
        if ins is not None:
            cur.index[ins.addr] = len(cur.insts)
            cur.insts.append(ins)
    for k in meta:
        ki = kernels.get(k[".name"])
        if ki is None:
            continue
        ki.args = k.get(".args", [])
        ki.kernarg_size = int(k.get(".kernarg_segment_size", 0))
        ki.lds_static = int(k.get(".group_segment_fixed_size", 0))
        ki.scratch = int(k.get(".private_segment_fixed_size", 0))
        ki.vgpr_count, ki.agpr_count, ki.sgpr_count = int(k.get(".vgpr_count", 0)), int(k.get(".agpr_count", 0)), int(k.get(".sgpr_count", 0))
        kd, _ = kds[k[".name"]]
        ki.rsrc3, ki.rsrc1, ki.rsrc2 = struct.unpack_from("<III", kd, 44)
        ki.code_props, ki.preload = struct.unpack_from("<HH", kd, 56)
        (entry_off,) = struct.unpack_from("<q", kd, 16)
        assert kds[k[".name"]][1] + entry_off == ki.entry, (k[".name"], hex(kds[k[".name"]][1] + entry_off), hex(ki.entry))
    return kernels


def load_code_object_file(path):
    """a stand-alone code object (hipcc --genco / --cuda-device-only -c, unbundled) -> {kernel symbol: KernelInfo}"""
    blob = open(path, "rb").read()
    if blob[:4] != b"\x7fELF":
        i = blob.find(b"\x7fELF")
        assert i >= 0, f"{path}: no ELF inside"
        # a clang offload bundle: take the gfx950 entry
        from tests.hipmock.kernargs import MAGIC
        if blob.startswith(MAGIC):
            (n,) = struct.unpack_from("<Q", blob, len(MAGIC))
            q = len(MAGIC) + 8
            for _ in range(n):
                off, size, idlen = struct.unpack_from("<QQQ", blob, q)
                ident = blob[q + 24:q + 24 + idlen].decode()
                q += 24 + idlen
                if "gfx950" in ident and size:
                    blob = blob[off:off + size]
                    break
        else:
            blob = blob[i:]
    return parse_code_object(blob)


def load_library(lib):
    """-> {kernel symbol: KernelInfo}; cached by the library's sha256"""
    blob = open(lib, "rb").read()
    key = hashlib.sha256(blob).hexdigest()[:16]
    os.makedirs(CACHE, exist_ok=True)
    cpath = os.path.join(CACHE, f"kernels_{key}_v3.pkl")
    if os.path.exists(cpath):
        with open(cpath, "rb") as f:
            return pickle.load(f)
    kernels = {}
    for co in code_objects(lib):
        kernels.update(parse_code_object(co))
    with open(cpath, "wb") as f:
        pickle.dump(kernels, f)
    return kernels


if __name__ == "__main__":
    from tests.gfx950sim import loader as _self        # (so that the cache pickles the package's classes, not __main__'s)
    ks = _self.load_library(sys.argv[1])
    for n, k in ks.items():
        print(f"{len(k.insts):6d} insts  lds={k.lds_static:6d} kernarg={k.kernarg_size:4d} rsrc2={k.rsrc2:#x} props={k.code_props:#x}  {n[:90]}")
