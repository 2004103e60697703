"""Independent check of the simulator's instruction semantics (tests/gfx950sim): probe kernels written with HIP-level operations
whose results the LANGUAGE defines -- __shfl_*, __ballot, integer and float division, conversions, 64-bit arithmetic, byte
permutes, LDS transposition with a barrier and atomics, and MFMA with the fragment layouts documented in
/opt/skills/guides/cdna_hip_programming.md section 3 -- are compiled here by hipcc for gfx950 and executed on the simulator;
whatever instructions the compiler picked (DPP, ds_bpermute, permlane swaps, SDWA, v_div_scale / fmas / fixup, v_rcp_iflag +
v_mul_hi corrections, v_perm, ...) must produce the defined result.  None of this repository's kernels is involved."""
import os
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump") or not shutil.which("objcopy"),
                                reason="needs hipcc and the ROCm LLVM tools")


class Arena:
    """'device' memory for stand-alone launches: one numpy buffer, bump-allocated"""

    def __init__(self, nbytes=1 << 24):
        self.buf = np.zeros(nbytes, np.uint8)
        self.base = self.buf.ctypes.data
        self.top = 0
        self.allocs = []

    def put(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.alloc(arr.nbytes)
        self.buf[p - self.base:p - self.base + arr.nbytes] = arr.view(np.uint8).ravel()
        return p

    def alloc(self, nbytes):
        p = self.base + self.top
        self.allocs.append((p, max(nbytes, 1)))
        self.top += (nbytes + 255) // 256 * 256
        assert self.top <= self.buf.size
        return p

    def get(self, p, dtype, shape):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return self.buf[p - self.base:p - self.base + n].view(dtype).reshape(shape).copy()


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    from tests.gfx950sim import loader
    from tests.gfx950sim.core import Memory
    from tests.gfx950sim.runtime import Simulator
    co = str(tmp_path_factory.mktemp("probe") / "probe.co")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "--cuda-device-only", "-c", os.path.join(HERE, "gfx950sim", "probe.hip"), "-o", co],
                   check=True, capture_output=True)
    # a Simulator without a library: the probe object's kernels, memory = the arena
    arena = Arena()
    s = Simulator(kernels=loader.load_code_object_file(co), mem=Memory(arena.base, arena.buf.size), strict=True, max_inst=5_000_000)
    s.arena = arena
    s.nproc = 1
    return s


def _launch(sim, name, grid, block, args, lds=0):
    sim.mem.set_allocs(sim.arena.allocs)
    raw = [a if isinstance(a, bytes) else int(a).to_bytes(8, "little") for a in args]
    L = sim.launch(name, grid, block, lds, raw)
    assert not L.hazards, L.hazards[:3]


def test_lane_operations(sim):
    rs = np.random.RandomState(0)
    v = rs.randint(-1000, 1000, 256).astype(np.int32)
    a = sim.arena
    pin, pout, p64 = a.put(v), a.alloc(40 * 256 * 4), a.alloc(2 * 256 * 8)
    _launch(sim, "probe_lanes", (1, 1, 1), (256, 1, 1), [pin, pout, p64])
    out = a.get(pout, np.int32, (40, 256))
    o64 = a.get(p64, np.uint64, (2, 256))
    w = v.reshape(4, 64)
    lane = np.arange(64)
    k = 0
    for m in (1, 2, 4, 8, 16, 32):
        assert np.array_equal(out[k].reshape(4, 64), w[:, lane ^ m]), f"__shfl_xor {m}"
        k += 1
    for d in (1, 3, 16):
        assert np.array_equal(out[k].reshape(4, 64), np.where(lane >= d, w[:, np.maximum(lane - d, 0)], w)), f"__shfl_up {d}"
        k += 1
    for d in (1, 5, 32):
        assert np.array_equal(out[k].reshape(4, 64), np.where(lane + d < 64, w[:, np.minimum(lane + d, 63)], w)), f"__shfl_down {d}"
        k += 1
    assert np.array_equal(out[k].reshape(4, 64), np.repeat(w[:, 17:18], 64, 1)); k += 1
    src = (np.arange(256) * 7 + 3) & 63
    assert np.array_equal(out[k], v.reshape(4, 64)[np.arange(256) // 64, src]); k += 1
    assert np.array_equal(out[k].reshape(4, 64), w[:, lane ^ 1]); k += 1                      # width 16, xor 1 stays inside
    assert np.array_equal(out[k].reshape(4, 64), np.where((lane & 7) >= 2, w[:, np.maximum(lane - 2, 0)], w)); k += 1
    ballots = [int(sum(1 << i for i in range(64) if w[q, i] & 1)) for q in range(4)]
    assert [int(x) for x in o64[0].reshape(4, 64)[:, 0]] == ballots
    assert np.array_equal(out[k].reshape(4, 64)[:, 0], [bin(b).count("1") for b in ballots]); k += 1
    assert np.array_equal(out[k].reshape(4, 64), np.tile(lane, (4, 1))); k += 1
    assert np.array_equal(out[k].reshape(4, 64), np.tile(lane, (4, 1))); k += 1               # mbcnt of a full mask = lane id
    assert np.array_equal(out[k].reshape(4, 64), np.repeat(w.sum(1, keepdims=True), 64, 1)); k += 1
    odd = [int(sum(1 << i for i in range(1, 64, 2) if w[q, i] > 0)) for q in range(4)]
    got = o64[1].reshape(4, 64)
    assert all(int(got[q, 1]) == odd[q] and int(got[q, 0]) == 0 for q in range(4))
    assert np.array_equal(out[k].reshape(4, 64), np.repeat(w[:, :1], 64, 1))


def test_integer_and_float_arithmetic(sim):
    rs = np.random.RandomState(1)
    n = 512
    ia = rs.randint(-2 ** 31, 2 ** 31 - 1, n).astype(np.int32)
    ib = rs.randint(-2 ** 31, 2 ** 31 - 1, n).astype(np.int32)
    ib[:64] = rs.randint(-50, 50, 64)
    ia[:16] = [0, 1, -1, 2 ** 31 - 1, -2 ** 31, 7, -7, 100, 65535, 65536, -65536, 3, 12345, -12345, 2, -2]
    ib[:16] = [1, -1, 1, 2, -1 if False else 3, -7, 7, 0, 65535, 65536, 255, 0, 1000, 1000, 2 ** 31 - 1, -2 ** 31]
    fa = (rs.standard_normal(n) * 10).astype(np.float32)
    fb = (rs.standard_normal(n) * 3).astype(np.float32)
    fb[np.abs(fb) < 1e-3] = 1.0
    a = sim.arena
    pi, pf, pd, pl = a.alloc(21 * n * 4), a.alloc(20 * n * 4), a.alloc(5 * n * 8), a.alloc(5 * n * 8)
    _launch(sim, "probe_arith", (2, 1, 1), (256, 1, 1), [a.put(ia), a.put(ib), a.put(fa), a.put(fb), pi, pf, pd, pl, int(n).to_bytes(4, "little")])
    io, fo = a.get(pi, np.int32, (21, n)), a.get(pf, np.float32, (20, n))
    do, lo = a.get(pd, np.float64, (5, n)), a.get(pl, np.uint64, (5, n))
    A, B = ia.astype(np.int64), ib.astype(np.int64)
    UA, UB = A & 0xFFFFFFFF, B & 0xFFFFFFFF
    nz = B != 0
    tdiv = np.where(nz, np.trunc(A / np.where(nz, B, 1)).astype(np.int64), 0)            # C division truncates (float64 is exact enough below 2^53)
    tdiv = np.where(nz, (np.abs(A) // np.where(nz, np.abs(B), 1)) * np.sign(A) * np.sign(B), 0)
    i32 = lambda x: ((np.asarray(x, np.int64) + 2 ** 31) % 2 ** 32 - 2 ** 31).astype(np.int32)
    sh = B & 31
    want = [i32(tdiv), i32(np.where(nz, A - tdiv * B, 0)), i32(np.where(UB != 0, UA // np.where(UB != 0, UB, 1), 0)),
            i32(np.where(UB != 0, UA % np.where(UB != 0, UB, 1), 0)), i32(A * B), i32((A * B) >> 32),
            i32((UA.astype(object) * UB.astype(object)) >> 32), i32(A >> sh), i32(UA >> sh), i32((UA << sh) & 0xFFFFFFFF),
            np.array([32 - int(x).bit_length() if x >= 0 else 0 for x in A], np.int32), np.array([bin(int(x)).count("1") for x in UA], np.int32),
            i32([int(f"{int(x):032b}"[::-1], 2) for x in UA]), i32(np.minimum(A, B)), i32(np.maximum(UA, UB)), i32(np.abs(A)),
            i32((A & 0xFFFF) * (B & 0xFFFF) + 7), i32(i32(A << 16) // 65536 + i32(B << 16) // 65536), i32(((A >> 8) & 0xFF) + ((B >> 16) & 0xFF))]
    for k, w in enumerate(want):
        assert np.array_equal(io[k], w), (k, io[k][:8], w[:8])

    def bperm(x, y, sel):
        src = (y.astype(np.uint64) << np.uint64(32)) | x.astype(np.uint64)
        out = np.zeros(n, np.uint64)
        for j in range(4):
            out |= ((src >> np.uint64(8 * ((sel >> (4 * j)) & 7))) & np.uint64(0xFF)) << np.uint64(8 * j)
        return i32(out.astype(np.int64))
    assert np.array_equal(io[19], bperm(UA, UB, 0x5140)) and np.array_equal(io[20], bperm(UA, UB, 0x3276))
    la, lb = (A << 20) + B, (B << 7) - A
    m64 = (1 << 64) - 1
    assert [int(x) for x in lo[0]] == [int(x) * int(y) & m64 for x, y in zip(la, lb)]
    assert [int(x) for x in lo[1]] == [(int(x) >> int(s & 63)) & m64 for x, s in zip(la, B)]
    assert [int(x) for x in lo[2]] == [(int(x) << int(s & 63)) & m64 for x, s in zip(la, A)]
    assert [int(x) for x in lo[3]] == [(int(x) + int(y)) & m64 for x, y in zip(la, lb)]
    assert [int(x) for x in lo[4]] == [(abs(int(x)) // abs(int(y)) * (1 if (x < 0) == (y < 0) else -1)) & m64 if y else 0 for x, y in zip(la, lb)]
    x, y = fa.astype(np.float64), fb.astype(np.float64)
    f32 = np.float32

    def close(got, want, ulps):
        want = np.asarray(want, np.float32)
        tol = ulps * np.spacing(np.abs(want).astype(np.float32)) + 1e-45
        assert (np.abs(got.astype(np.float64) - want.astype(np.float64)) <= tol).all(), (got[:6], want[:6])
    close(fo[0], f32(x / y), 0)                                   # IEEE division: correctly rounded
    close(fo[1], np.sqrt(np.abs(fa)), 1)
    close(fo[2], f32(x * y + 0.25), 0)
    close(fo[3], np.exp2(f32(fa * f32(0.125)).astype(np.float64)), 2)
    close(fo[4], np.exp(f32(fa * f32(0.125)).astype(np.float64)), 4)
    close(fo[5], 1.0 / np.sqrt(np.abs(y) + 1.0), 2)
    close(fo[6], np.minimum(fa, fb), 0)
    close(fo[7], np.maximum(np.maximum(fa, fb), f32(0.5)), 0)
    t37 = (fa * f32(3.7)).astype(np.float32)
    close(fo[8], np.floor(t37), 0)
    close(fo[9], np.rint(t37), 0)
    close(fo[10], np.trunc(t37), 0)
    close(fo[11], ia.astype(np.float32), 0)
    close(fo[12], UA.astype(np.float32), 0)
    close(fo[13], np.trunc((fa * f32(1000.0)).astype(np.float32)).astype(np.float32), 0)
    close(fo[14], np.trunc((np.abs(fa) * f32(1000.0)).astype(np.float32)).astype(np.float32), 0)
    close(fo[15], fa.astype(np.float16).astype(np.float32), 0)
    u = fa.view(np.uint32).astype(np.uint64)
    bf = (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)
    close(fo[16], bf, 0)
    close(fo[17], (fa.astype(np.float16) + fb.astype(np.float16)).astype(np.float32), 0)
    close(fo[18], np.ldexp(fa, (ib & 7)), 0)
    close(fo[19], np.where(fa < fb, fa, np.where(fa == fb, 0, fb)), 0)
    assert np.allclose(do[0], x / y, rtol=1e-15) and np.allclose(do[1], np.sqrt(np.abs(x) + 1), rtol=1e-15)
    assert np.allclose(do[2], x * y + 0.5, rtol=1e-15) and np.allclose(do[3], 1 / np.sqrt(np.abs(y) + 2), rtol=1e-15)
    assert np.array_equal(do[4], A * 0.5)


def test_mfma_fragment_layouts(sim):
    """asymmetric operands: a row / column swap or a wrong lane mapping in the simulator's MFMA cannot pass"""
    rs = np.random.RandomState(2)
    f16 = np.float16
    A32, B32, C32 = (rs.standard_normal((32, 16)) * 0.5).astype(f16), (rs.standard_normal((16, 32)) * 0.5).astype(f16), rs.standard_normal((32, 32)).astype(np.float32)
    A16, B16 = (rs.standard_normal((16, 32)) * 0.5).astype(f16), (rs.standard_normal((32, 16)) * 0.5).astype(f16)
    Af, Bf = rs.standard_normal((32, 4)).astype(np.float32), rs.standard_normal((4, 32)).astype(np.float32)
    Ab = (rs.standard_normal((32, 16)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    Bb = (rs.standard_normal((16, 32)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    a = sim.arena
    pD32, pD16, pDf32, pDf16, pDb = (a.alloc(32 * 32 * 4) for _ in range(5))
    _launch(sim, "probe_mfma", (1, 1, 1), (64, 1, 1), [a.put(A32), a.put(B32), a.put(C32), pD32, a.put(A16), a.put(B16), pD16, a.put(Af), a.put(Bf),
                                                       pDf32, pDf16, a.put(Ab), a.put(Bb), pDb])
    D32 = a.get(pD32, np.float32, (32, 32))
    assert np.allclose(D32, A32.astype(np.float64) @ B32.astype(np.float64) + C32, rtol=1e-6, atol=1e-6)
    D16 = a.get(pD16, np.float32, (16, 16))
    assert np.allclose(D16, A16.astype(np.float64) @ B16.astype(np.float64), rtol=1e-6, atol=1e-6)
    assert np.allclose(a.get(pDf32, np.float32, (32, 32)), Af.astype(np.float64) @ Bf.astype(np.float64), rtol=1e-6, atol=1e-6)
    assert np.allclose(a.get(pDf16, np.float32, (16, 16)), Af[:16].astype(np.float64) @ Bf[:, :16].astype(np.float64), rtol=1e-6, atol=1e-6)
    bf = lambda u: (u.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    assert np.allclose(a.get(pDb, np.float32, (32, 32)), bf(Ab) @ bf(Bb), rtol=1e-6, atol=1e-6)


def test_lds_barrier_and_atomics(sim):
    rs = np.random.RandomState(3)
    x = rs.standard_normal((32, 32)).astype(np.float32)
    a = sim.arena
    pout, ph = a.alloc(1024 * 4 + 512), a.alloc(64)
    _launch(sim, "probe_lds", (1, 1, 1), (256, 1, 1), [a.put(x), pout, ph])
    t = np.arange(256)
    hs = ((t * 257) & 0xFFFF).astype(np.uint16)
    out = a.get(pout, np.float32, (32, 32))
    want = x.T.copy()
    for r in range(0, 32, 8):
        for tt in t:
            want[(tt >> 5) + r, tt & 31] += float(hs[255 - tt])
    assert np.array_equal(out, want.astype(np.float32))
    hist = a.get(ph, np.int32, (16,))
    assert np.array_equal(hist, np.bincount((t * 7) & 15, weights=t, minlength=16).astype(np.int32))
    tail = a.buf[pout - a.base + 4096:pout - a.base + 4096 + 512].view(np.uint16)
    assert np.array_equal(tail, hs)


def test_hazard_recogniser_fires_on_deliberately_wrong_kernels(sim):
    """two probes that are WRONG on purpose: a register consumed before any s_waitcnt covers its load, and an LDS-DMA destination read
    back without the wave's vmcnt wait.  Hardware would compute with stale data (sometimes); the simulator must say so every time."""
    a = sim.arena
    v = np.arange(64, dtype=np.int32)
    sim.strict = False
    try:
        sim.mem.set_allocs(a.allocs)
        pin, pout = a.put(v), a.alloc(256)
        sim.mem.set_allocs(a.allocs)
        L = sim.launch("probe_missing_wait", (1, 1, 1), (64, 1, 1), 0, [int(pin).to_bytes(8, "little"), int(pout).to_bytes(8, "little")])
        assert any("while a load into it is in flight" in h for h in L.hazards), L.hazards
        x = np.arange(256, dtype=np.float32)
        pin, pout = a.put(x), a.alloc(1024)
        sim.mem.set_allocs(a.allocs)
        L = sim.launch("probe_lds_dma_race", (1, 1, 1), (64, 1, 1), 0, [int(pin).to_bytes(8, "little"), int(pout).to_bytes(8, "little")])
        assert any("LDS-DMA" in h for h in L.hazards), L.hazards
        # and the stale data is what was read: zeros, not the values the DMA delivered afterwards
        assert not np.array_equal(a.get(pout, np.float32, (64,)), x[:64])
    finally:
        sim.strict = True
