"""TEST INFRASTRUCTURE: wave / workgroup state, operand access and the memory model of the gfx950 instruction-level simulator.

What is modelled (and why): the simulator exists to execute the SHIPPED machine code of libbndm_hip.so on a machine without
a GPU, so that device code -- not a restatement of it -- is compared with oracle/.  Functional semantics only, plus the one
timing-related property hand-written kernels get wrong: **every memory operation completes as late as the ISA allows**.
A VMEM load's destination registers (or, for LDS-DMA, the LDS bytes) change only when an `s_waitcnt vmcnt(N)` of the issuing
wave requires it (in issue order); LDS reads return at the covering `lgkmcnt`; LDS writes become visible to OTHER waves only
then (the issuing wave's own later LDS operations see them: the LDS pipeline is in order).  Reading a register or an LDS
byte whose load is still in flight is recorded as a hazard.  So a kernel whose counted waits are one too loose computes with
stale data here, deterministically, where hardware would do so only under load.

Not modelled: cycle counts, bank conflicts, caches (memory is coherent), MFMA / VALU data hazards (s_nop distances),
trap handlers, scratch memory, wave32."""
import ctypes as C
import re
import struct

import numpy as np

np.seterr(all="ignore")

U8, U16, U32, U64 = np.uint8, np.uint16, np.uint32, np.uint64
I16, I32, I64 = np.int16, np.int32, np.int64
F16, F32, F64 = np.float16, np.float32, np.float64
M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF
LANE = np.arange(64, dtype=np.int64)
VCC, M0, EXEC = 106, 124, 126
ACC0 = 256                       # a0.. live behind v0..v255 in the same array (one file on gfx950)

BARRIER, ENDPGM = -1, -2


class SimError(Exception):
    pass


def sx(x, bits):
    x &= (1 << bits) - 1
    return x - (1 << bits) if x >> (bits - 1) else x


def mask_to_bool(m):
    return np.unpackbits(np.frombuffer(int(m & M64).to_bytes(8, "little"), U8), bitorder="little").astype(bool)


def bool_to_mask(b):
    return int.from_bytes(np.packbits(b, bitorder="little").tobytes(), "little")


def full(x):
    return np.full(64, x & M32, U32)


def full64(x):
    return np.full(64, x & M64, U64)


# ---------------------------------------------------------------------------------------------------------------------
# device memory: the recording runtime's "device" region is ordinary memory of this process at a fixed address
# ---------------------------------------------------------------------------------------------------------------------
class Memory:
    def __init__(self, base=0x200000000000, span=1 << 40):
        self.base, self.span = base, span
        self.u8 = np.ctypeslib.as_array(C.cast(C.c_void_p(base), C.POINTER(C.c_uint8)), shape=(span,))
        self.allocs = None               # sorted (starts, ends) of live allocations, or None: no checking
        self.faults = []
        self.extra = []                  # (addr, numpy uint8 array) host-side segments (kernarg)
        self.undo = None                 # list of (byte indices, old bytes) while a launch is run differentially
        self.counters = {}               # bytes moved by vector memory instructions ("load", "store"), per launch (GFX950SIM_STATS)

    def set_allocs(self, pairs):
        pairs = sorted(pairs)
        self.allocs = (np.array([p for p, _ in pairs], np.int64), np.array([p + n for p, n in pairs], np.int64))

    def check(self, addr, nbytes, what, active=None):
        """addr: int64 array of byte addresses; every active access must lie inside one live allocation"""
        if self.allocs is None:
            return
        a = addr if active is None else addr[active]
        if a.size == 0:
            return
        starts, ends = self.allocs
        i = np.searchsorted(starts, a, side="right") - 1
        ok = (i >= 0) & (a + nbytes <= ends[np.maximum(i, 0)])
        if not ok.all():
            bad = a[~ok][0]
            self.faults.append(f"{what}: address {int(bad):#x} (+{nbytes}) outside every live allocation")
            raise SimError(self.faults[-1])

    def host_read(self, addr, n):
        for a0, arr in self.extra:
            if a0 <= addr and addr + n <= a0 + arr.size:
                return bytes(arr[addr - a0:addr - a0 + n])
        if self.base <= addr < self.base + self.span:
            if self.allocs is not None:
                self.check(np.array([addr], np.int64), n, "scalar load")
            return bytes(self.u8[addr - self.base:addr - self.base + n])
        raise SimError(f"scalar load from {addr:#x}: not device memory and not the kernarg segment")

    def gather(self, addr, ndw, active):
        """addr int64[64] byte addresses (4-aligned), -> uint32 [ndw, 64]; inactive lanes read 0"""
        off = addr - self.base
        out = np.zeros((ndw, 64), U32)
        if not active.any():
            return out
        a = off[active]
        if (a & 3).any():
            raise SimError("misaligned dword access")
        self.counters["load"] = self.counters.get("load", 0) + 4 * ndw * a.size
        idx = a[None, :] + (4 * np.arange(ndw, dtype=np.int64))[:, None]            # [ndw, n]
        b = self.u8[(idx[:, :, None] + np.arange(4, dtype=np.int64)).reshape(-1)].reshape(ndw, -1, 4)
        out[:, active] = b.view(U32).reshape(ndw, -1) if b.flags.c_contiguous else np.ascontiguousarray(b).view(U32).reshape(ndw, -1)
        return out

    def scatter(self, addr, data, active):
        """data uint32 [ndw, 64]"""
        off = addr - self.base
        if not active.any():
            return
        a = off[active]
        ndw = data.shape[0]
        self.counters["store"] = self.counters.get("store", 0) + 4 * ndw * a.size
        idx = a[None, :] + (4 * np.arange(ndw, dtype=np.int64))[:, None]
        b = np.ascontiguousarray(data[:, active]).view(U8).reshape(ndw, -1, 4)
        flat = (idx[:, :, None] + np.arange(4, dtype=np.int64)).reshape(-1)
        if self.undo is not None:
            self.undo.append((flat, self.u8[flat].copy()))
        self.u8[flat] = b.reshape(-1)

    def gather_small(self, addr, nbytes, active):
        """sub-dword loads: -> uint32[64] zero-extended"""
        off = addr - self.base
        out = np.zeros(64, U32)
        a = off[active]
        v = np.zeros(a.size, U32)
        for k in range(nbytes):
            v |= self.u8[a + k].astype(U32) << U32(8 * k)
        out[active] = v
        return out

    def scatter_small(self, addr, val, nbytes, active):
        a = (addr - self.base)[active]
        v = val[active]
        for k in range(nbytes):
            if self.undo is not None:
                self.undo.append((a + k, self.u8[a + k].copy()))
            self.u8[a + k] = ((v >> U32(8 * k)) & U32(0xFF)).astype(U8)


# ---------------------------------------------------------------------------------------------------------------------
# wave and workgroup state
# ---------------------------------------------------------------------------------------------------------------------
class Pending:
    """one memory operation in flight: apply() makes its effect visible"""
    __slots__ = ("apply", "regs", "sregs", "lds", "what")

    def __init__(self, apply, regs=(), sregs=(), lds=None, what=""):
        self.apply, self.regs, self.sregs, self.lds, self.what = apply, regs, sregs, lds, what


class Workgroup:
    def __init__(self, launch, wg_id, nwaves):
        self.launch = launch
        self.id = wg_id
        self.lds_size = launch.lds_bytes
        n = max(self.lds_size, 4)
        self.lds = np.zeros((n + 3) // 4 * 4, U8)
        self.lds32 = self.lds.view(U32)
        self.lds_pending = np.zeros(self.lds.size, np.int16)     # LDS-DMA bytes in flight (count per byte)
        self.lds_owner = np.full(self.lds.size, -1, np.int16)    # ... and the wave that issued the latest one (diagnostics)
        # bytes whose content is undefined: a store (or a second DMA) raced an LDS-DMA in flight.  Writing such bytes is not
        # an error (kernels park piece-less DMAs and padding write-backs in a dead slot); READING them is.
        self.lds_taint = np.zeros(self.lds.size, bool)
        self.waves = []
        self.hazards = launch.hazards


class Wave:
    def __init__(self, wg, wid, kernel):
        self.wg = wg
        self.wid = wid
        self.kernel = kernel
        self.s = [0] * 128
        self.v = np.zeros((512, 64), U32)
        self.scc = 0
        self.pc = 0
        self.execb = np.ones(64, bool)
        self.exec_full = True
        self.vmq = []                        # VMEM operations in flight, oldest first
        self.lgq = []                        # LDS / SMEM operations in flight, oldest first
        self.vpend = np.zeros(512, np.int16)
        self.spend = [0] * 128
        self.npend = 0                       # registers with a load in flight (fast path: 0)
        self.own_lds_writes = 0              # LDS writes of this wave not yet applied
        self.done = False
        self.ninst = 0
        self.clock = 1000 + 17 * wid
        self.mem = wg.launch.mem
        self.stats_on = wg.launch.stats is not None          # GFX950SIM_STATS=1: instruction counts, bytes moved, LDS bank cycles
        nscr = (kernel.scratch + 3) // 4
        self.scratch = np.zeros((nscr, 64), U32) if nscr else None      # private segment: register spill slots, per lane

    # --- exec -----------------------------------------------------------------------------------------------------
    def set_exec(self, m):
        m &= M64
        self.s[EXEC], self.s[EXEC + 1] = m & M32, m >> 32
        self.execb = mask_to_bool(m)
        self.exec_full = m == M64

    def get_exec(self):
        return self.s[EXEC] | (self.s[EXEC + 1] << 32)

    def sync_exec(self):
        self.set_exec(self.get_exec())

    # --- hazards ---------------------------------------------------------------------------------------------------
    def hazard(self, text):
        ins = self.kernel.insts[self.pc]
        msg = f"{text} at {ins.addr:#x} `{ins.text}` (wave {self.wid}, workgroup {self.wg.id})"
        hz = self.wg.hazards
        if len(hz) < 200:
            hz.append(msg)
        if self.wg.launch.strict:
            raise SimError(msg)

    # --- register writes -------------------------------------------------------------------------------------------
    def wv(self, r, val):
        if self.npend and self.vpend[r]:
            self.hazard(f"write of v{r} while a load into it is in flight")
        if self.exec_full:
            self.v[r] = val
        else:
            np.copyto(self.v[r], val, where=self.execb)

    def wv_all(self, r, val):
        """write ignoring exec (MFMA results, load returns carry their own mask)"""
        self.v[r] = val

    def wv64(self, r, val64):
        self.wv(r, (val64 & U64(M32)).astype(U32))
        self.wv(r + 1, (val64 >> U64(32)).astype(U32))

    def ws(self, r, val):
        if self.npend and self.spend[r]:
            self.hazard(f"write of s{r} while a load into it is in flight")
        self.s[r] = val & M32
        if r >= EXEC:
            self.sync_exec()

    def ws64(self, r, val):
        if self.npend and (self.spend[r] or self.spend[r + 1]):
            self.hazard(f"write of s[{r}:{r + 1}] while a load into it is in flight")
        self.s[r], self.s[r + 1] = val & M32, (val >> 32) & M32
        if r >= EXEC:
            self.sync_exec()

    # --- in-flight memory operations ---------------------------------------------------------------------------------
    def push_vm(self, p):
        for r in p.regs:
            self.vpend[r] += 1
        self.npend += len(p.regs)
        self.vmq.append(p)

    def push_lgkm(self, p):
        for r in p.regs:
            self.vpend[r] += 1
        for r in p.sregs:
            self.spend[r] += 1
        self.npend += len(p.regs) + len(p.sregs)
        self.lgq.append(p)

    def _retire(self, p):
        for r in p.regs:
            self.vpend[r] -= 1
        for r in p.sregs:
            self.spend[r] -= 1
        self.npend -= len(p.regs) + len(p.sregs)
        p.apply()

    def wait_vm(self, n):
        q = self.vmq
        while len(q) > n:
            self._retire(q.pop(0))

    def wait_lgkm(self, n):
        q = self.lgq
        while len(q) > n:
            self._retire(q.pop(0))

    def flush_own_lds_writes(self):
        """the LDS pipeline is in order per wave: a later LDS operation of this wave sees its earlier writes.  Their queue
        entries stay (they still count in lgkmcnt) but their effect is applied now."""
        if self.own_lds_writes:
            for p in self.lgq:
                if p.lds is not None and p.apply is not _noop:
                    p.apply()
                    p.apply = _noop
            self.own_lds_writes = 0

    def drain(self):
        self.wait_vm(0)
        self.wait_lgkm(0)


def _noop():
    pass


# ---------------------------------------------------------------------------------------------------------------------
# operand parsing
# ---------------------------------------------------------------------------------------------------------------------
_SPECIAL = {"vcc": ("s", VCC, 2), "vcc_lo": ("s", VCC, 1), "vcc_hi": ("s", VCC + 1, 1), "exec": ("s", EXEC, 2),
            "exec_lo": ("s", EXEC, 1), "exec_hi": ("s", EXEC + 1, 1), "m0": ("s", M0, 1),
            "flat_scratch": ("s", 102, 2), "flat_scratch_lo": ("s", 102, 1), "flat_scratch_hi": ("s", 103, 1),
            "xnack_mask": ("s", 104, 2)}
_RE1 = re.compile(r"([vsa])(\d+)$")
_RE2 = re.compile(r"([vsa])\[(\d+):(\d+)\]$")


def parse_reg(tok):
    """-> (file 'v'|'s', first index, count) or None.  AGPRs map to the 'v' file at +256."""
    t = _SPECIAL.get(tok)
    if t:
        return t
    m = _RE1.match(tok)
    if m:
        f, i = m.group(1), int(m.group(2))
        return ("v", i + ACC0, 1) if f == "a" else (f, i, 1)
    m = _RE2.match(tok)
    if m:
        f, i, j = m.group(1), int(m.group(2)), int(m.group(3))
        return ("v", i + ACC0, j - i + 1) if f == "a" else (f, i, j - i + 1)
    return None


class Opnd:
    """a parsed source / destination operand"""
    __slots__ = ("text", "reg", "neg", "abs", "sext", "const")

    def __init__(self, text):
        t = text.strip()
        self.text = t
        self.neg = self.abs = self.sext = False
        if t.startswith("sext(") and t.endswith(")"):
            self.sext, t = True, t[5:-1]
        if t.startswith("-") and len(t) > 1 and (t[1] in "vsa|" or t[1:] in _SPECIAL):
            self.neg, t = True, t[1:]
        if t.startswith("neg(") and t.endswith(")"):
            self.neg, t = True, t[4:-1]
        if t.startswith("|") and t.endswith("|"):
            self.abs, t = True, t[1:-1]
        if t.startswith("abs(") and t.endswith(")"):
            self.abs, t = True, t[4:-1]
        self.reg = parse_reg(t)
        self.const = None if self.reg else t

    def is_reg(self):
        return self.reg is not None


_APERTURE = {"src_shared_base": 0x0001000000000000, "src_private_base": 0x0002000000000000,
             "src_shared_limit": 0x000100000003FFFF, "src_private_limit": 0x00020000FFFFFFFF}
_FLOAT_RE = re.compile(r"^-?\d+\.\d*(e[-+]?\d+)?$|^-?\d+e[-+]?\d+$", re.I)


def const_bits(tok, kind):
    """literal / inline constant -> integer bit pattern for an operand of `kind` ('i32','i64','f32','f16','f64','u64')"""
    if tok == "scc":
        raise SimError("scc as a constant")
    if tok in ("off", "null"):
        return 0
    if tok in _APERTURE:                 # flat-address apertures (only compared / subtracted by compiled code, never dereferenced here)
        return _APERTURE[tok] if kind in ("i64", "u64", "f64") else _APERTURE[tok] >> 32
    if _FLOAT_RE.match(tok):
        x = float(tok)
        if tok in ("0.15915494", "0.15915494309189532"):
            x = 0.15915494309189532
        if kind == "f16":
            return int(np.array([x], F16).view(U16)[0])
        if kind in ("f64", "i64", "u64"):
            return struct.unpack("<Q", struct.pack("<d", x))[0]
        return struct.unpack("<I", struct.pack("<f", x))[0]
    v = int(tok, 0)
    if kind == "f64":
        # 32-bit literal of a 64-bit float operand: the HIGH dword; small inline integers are integers
        if tok.lower().startswith("0x") or abs(v) > 64:
            return (v & M32) << 32
        return v & M64
    if kind in ("i64",):
        return v & M64 if v < 0 else v          # negative inline constants sign-extend
    if kind == "u64":
        return v & M64
    return v & M32
