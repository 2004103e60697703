"""TEST INFRASTRUCTURE: cases that drive libbndm_hip.so through its C ABI (ctypes + numpy, no torch device) under the recording
HIP stand-in WITH the instruction-level simulator installed, and compare what the library's machine code computed with
oracle/ -- the same comparisons tests/test_gpu_steps.py and tests/test_gpu_noise.py make on a real MI355X.

    LD_LIBRARY_PATH=<stand-in dir> HIPMOCK_TRACE=t.txt HIPMOCK_KERNARGS=ka.txt python tests/gfx950sim/cases.py <lib.so> <case> ...

Prints one JSON line per case: {"case":, "ok":, "launches":, "wave_instructions":, "hazards":, ...}.  (UNet / VAE / sampling
loop cases run through tests/hipmock/exec_forward.py with EXEC_SIM=1, which shares its set-up with the host-side replay.)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bndm_amd import _lib  # noqa: E402
from tests.hipmock import drive  # noqa: E402
from tests.hipmock.check_conv_t32 import dev  # noqa: E402
from tests.gfx950sim.runtime import Simulator  # noqa: E402


class Ctx:
    def __init__(self, libpath):
        _lib.LIB_PATH = libpath
        self.lib = _lib.load()
        self.d = drive.Dev()
        self.sim = Simulator(libpath, verbose=os.environ.get("SIM_VERBOSE") == "1").install()

    def up(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.d.alloc(max(arr.nbytes, 16))
        dev(p.value, arr.dtype, arr.size)[:] = arr.ravel()
        return p

    def alloc(self, nbytes):
        return self.d.alloc(max(nbytes, 16))

    def down(self, p, dtype, shape):
        return dev(p.value, dtype, int(np.prod(shape))).reshape(shape).copy()


def case_steps(cx):
    """iadb_step (both channel forms), ddim_step, export_u8 (both roundings), train targets: bit-exact vs the oracle's fp32
    operation order (tests/test_gpu_steps.py)"""
    import torch
    from oracle import sampler_oracle as SO
    lib, rs = cx.lib, np.random.RandomState(3)
    out = {}
    B, Cc, HW = 3, 3, 320                                         # ragged: not a multiple of the 4-wide vector loop x 256 threads
    x = rs.standard_normal((B, Cc, HW)).astype(np.float32)
    for co, (da, dg) in ((6, (-0.004, -0.0031)), (3, (-0.004, 0.0))):
        d = rs.standard_normal((B, co, HW)).astype(np.float32)
        px, pd = cx.up(x), cx.up(d)
        _lib.check(lib.bndm_iadb_step(px, pd, da, dg, B, Cc, co, HW, None), "iadb_step")
        got = cx.down(px, np.float32, x.shape)
        want = x + np.float32(da) * d[:, :Cc]
        if co == 2 * Cc:
            want = want + np.float32(dg) * d[:, Cc:]
        out[f"iadb_step_cout{co}_bitexact"] = bool(np.array_equal(got, want))
    eps = rs.standard_normal(x.shape).astype(np.float32)
    px, pe = cx.up(3 * x), cx.up(eps)
    sat, s1at, sap, s1ap = (np.float32(v) for v in (0.9, 0.43588989, 0.92, 0.39191836))
    _lib.check(lib.bndm_ddim_step(px, pe, sat, s1at, sap, s1ap, 1.0, x.size, None), "ddim_step")
    got = cx.down(px, np.float32, x.shape)
    x0 = np.clip((3 * x - s1at * eps) / sat, np.float32(-1), np.float32(1))          # ddim_diffusers.py:680 (eps-prediction, eta 0)
    want = sap * x0 + s1ap * eps
    out["ddim_step_rel"] = float(np.abs(got - want).max() / np.abs(want).max())
    px = cx.up(3 * x)                                           # clip <= 0: no clipping of the predicted x0
    _lib.check(lib.bndm_ddim_step(px, pe, sat, s1at, sap, s1ap, 0.0, x.size, None), "ddim_step")
    want = sap * ((3 * x - s1at * eps) / sat) + s1ap * eps
    out["ddim_step_noclip_rel"] = float(np.abs(cx.down(px, np.float32, x.shape) - want).max() / np.abs(want).max())
    img = (rs.standard_normal((2, 3, 100)) * 0.7).astype(np.float32)
    for rnd in (0, 1):
        po = cx.alloc(img.size)
        _lib.check(lib.bndm_export_u8(cx.up(img), po, 2, 3, 100, rnd, None), "export_u8")
        got = cx.down(po, np.uint8, (2, 100, 3))
        want = SO.export_u8(torch.from_numpy(img.reshape(2, 3, 10, 10)), rounding="round" if rnd else "trunc").reshape(2, 100, 3)
        out[f"export_u8_round{rnd}_bitexact"] = bool(np.array_equal(got, np.asarray(want)))
    # training-time noise injection: forward blend + regression targets, bit-exact vs the oracle's operation order
    Bt, per = 3, 4 * 9 * 9                                   # ragged: 324 elements per sample
    t = [rs.standard_normal((Bt, 4, 9, 9)).astype(np.float32) for _ in range(4)]
    al, alp = rs.rand(Bt).astype(np.float32), rs.rand(Bt).astype(np.float32)
    po = [cx.alloc(Bt * per * 4) for _ in range(4)]
    _lib.check(lib.bndm_iadb_train_targets(cx.up(t[0]), cx.up(t[1]), cx.up(t[2]), cx.up(t[3]), cx.up(al), cx.up(alp), *po, Bt, per, None), "train_targets")
    want = SO.train_targets(*(torch.from_numpy(v) for v in t), torch.from_numpy(al), torch.from_numpy(alp))
    for nm, p_, w_ in zip(("x_alpha", "tar1", "tar2", "tar"), po, want):
        out[f"train_{nm}_bitexact"] = bool(np.array_equal(cx.down(p_, np.float32, t[0].shape), w_.numpy()))
    ok = all(v is True or (isinstance(v, float) and v <= 1e-6) for v in out.values())
    return ok, out


def _noise(cx, L_host, pL, z, alpha, res, mode, layout, B, b0, bn, dense=0):
    Cc = z.shape[1]
    shape = (bn, Cc, res, res)
    n = int(np.prod(shape)) * 4
    o1, o2, o3 = cx.alloc(n), cx.alloc(n), cx.alloc(n)
    ws = cx.lib.bndm_bluenoise_workspace_bytes(bn, Cc, res)
    pw = cx.alloc(ws)
    pa = cx.up(alpha) if alpha is not None else None
    _lib.check(cx.lib.bndm_bluenoise(pL, dense, cx.up(z), layout, pa, o1, o2, o3, B, b0, bn, Cc, res, mode, pw, ws, None), "bluenoise")
    return [cx.down(p, np.float32, shape) for p in (o1, o2, o3)]


def case_noise(cx, which):
    """bndm_bluenoise vs oracle/noise_oracle.py (itself pinned by the reference's goldens): max-abs <= 1e-4 * max|ref|
    (tests/test_gpu_noise.py's bar)"""
    from bndm_amd.synth import formula_factor
    from oracle import noise_oracle as NO
    Lh = formula_factor()
    pL = cx.up(Lh)
    rs = np.random.RandomState(1)
    out, worst = {}, 0.0

    def cmp(tag, got, ref):
        nonlocal worst
        for g, r, nm in zip(got, ref, ("noise", "bn", "wn")):
            e = float(np.abs(g - r).max() / max(1.0, np.abs(r).max()))
            out[f"{tag}_{nm}"] = e
            worst = max(worst, e)

    if which == "small64":          # n = 6 columns: bluenoise_small<W16> + finish (the HBM regime)
        z = rs.standard_normal((2, 3, 64, 64)).astype(np.float32)
        a = np.array([0.25, 1.0], np.float32)
        cmp("blend", _noise(cx, Lh, pL, z, a, 64, 0, 0, 2, 0, 2), NO.get_noise_v2(z, Lh, a, "gaussianBN", "test"))
    elif which == "small32col":     # 24 columns: bluenoise_small<W32>, 32-px crop, GBN
        x = rs.standard_normal((6, 4, 32, 32)).astype(np.float32)
        cmp("gbn32", _noise(cx, Lh, pL, x, None, 32, 1, 1, 6, 0, 6), NO.get_noise_v2(x, Lh, None, "GBN", "test"))
    elif which == "gemm128":        # 128 px, B = 4: 48 columns -> bluenoise_gemm, tile permutation + scrambled noise_wn
        x = rs.standard_normal((4, 3, 128, 128)).astype(np.float32)
        a = np.linspace(0.1, 0.9, 4).astype(np.float32)
        cmp("blend128", _noise(cx, Lh, pL, x, a, 128, 0, 2, 4, 0, 4), NO.get_noise_v2(x, Lh, a, "gaussianBN", "test"))
        # a shard of the same global batch (SURVEY 8e caveat 1): samples 1..2 of 4 equal the full result's rows
        full = NO.get_noise_v2(x, Lh, a, "gaussianBN", "test")
        cmp("shard", _noise(cx, Lh, pL, x, a, 128, 0, 2, 4, 1, 2), [f[1:3] for f in full])
    elif which in ("gemm64_b22", "gemm64_b64"):     # 64 px at 66 / 192 columns: bluenoise_gemm<NT=2> / <NT=3> (c2's own call: B = 64)
        Bn = 22 if which.endswith("b22") else 64
        z = rs.standard_normal((Bn, 3, 64, 64)).astype(np.float32)
        a = np.linspace(0.0, 1.0, Bn).astype(np.float32)
        cmp("blend", _noise(cx, Lh, pL, z, a, 64, 0, 0, Bn, 0, Bn), NO.get_noise_v2(z, Lh, a, "gaussianBN", "test"))
    elif which == "dense64":        # l_dense = 1: the reference's semantics for an arbitrary matrix
        z = rs.standard_normal((2, 3, 64, 64)).astype(np.float32)
        a = np.array([0.5, 0.75], np.float32)
        cmp("dense", _noise(cx, Lh, pL, z, a, 64, 0, 0, 2, 0, 2, dense=1), NO.get_noise_v2(z, Lh, a, "gaussianBN", "test"))
    else:
        raise SystemExit(f"unknown noise case {which}")
    return worst <= 1e-4, out


def main():
    libpath = os.path.abspath(sys.argv[1])
    cx = Ctx(libpath)
    rc = 0
    for case in sys.argv[2:]:
        n0, t0 = len(cx.sim.log), time.time()
        if case == "steps":
            ok, info = case_steps(cx)
        elif case.startswith("noise:"):
            ok, info = case_noise(cx, case.split(":", 1)[1])
        else:
            raise SystemExit(f"unknown case {case}")
        log = cx.sim.log[n0:]
        ok = ok and not cx.sim.hazards
        print(json.dumps(dict(case=case, ok=bool(ok), launches=len(log), wave_instructions=sum(l[2] for l in log), hazards=cx.sim.hazards[:5],
                              seconds=round(time.time() - t0, 1), kernels=sorted({l[0].split("EE")[0][-40:] for l in log}), **info)), flush=True)
        rc |= 0 if ok else 1
    cx.sim.uninstall()
    sys.exit(rc)


if __name__ == "__main__":
    main()
