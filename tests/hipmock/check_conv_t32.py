"""TEST INFRASTRUCTURE: what the engine asks conv_t32 to compute, checked on the CPU against the layers' definition.

Runs under the recording HIP runtime (tests/hipmock/hipmock.cpp: "device" memory is host memory, launches are recorded, nothing
executes).  For every conv_t32 launch of one UNet forward (res64, 3 -> 6, B = 1):

  * the sources the launch points at are filled with random 16-bit data, the GroupNorm partial-sum buffers with the sums of that
    data, the additive row (bias or time-embedding row) and the residual with random values;
  * MODEL: the launch is evaluated in numpy exactly as its kernel arguments and uploaded tables describe it -- segments in
    FusedArgs order (concatenation, nearest-2x sources, raw 1x1 shortcut segments), GroupNorm finalised from the partial sums
    with the uploaded gamma / beta, weights DECODED from the packed, pre-swizzled upload (pack_weights_t32: [n-tile][K-step]
    [row][slot ^ ((row >> 2) & 3)][8]), additive row, residual, 16-bit roundings where the kernel rounds;
  * REFERENCE: the same layer from the state dict's ORIGINAL tensors by the module's definition (diffusers ResnetBlock2D /
    Upsample2D / conv_norm_out + conv_out as restated in oracle/unet_oracle.py), fp32.

Agreement (rel-L2 <= 5e-3: the 16-bit roundings) says that weight packing, K-step order, the ln 2 fold, segment order and
offsets, gamma / beta / bias wiring and the buffer dataflow of the launch list are right -- everything about the dominant kernel
that is decided on the host.  It says nothing about the kernel's device code.

    LD_LIBRARY_PATH=<stand-in dir> HIPMOCK_TRACE=t.txt HIPMOCK_KERNARGS=ka.txt python tests/hipmock/check_conv_t32.py lib.so
"""
import ctypes as C
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from bndm_amd import _lib  # noqa: E402
from tests.hipmock import drive, harness as H  # noqa: E402

LOG2E, LN2 = 1.4426950408889634, 0.6931471805599453


class Seg(C.Structure):
    _fields_ = [("src", C.c_uint64), ("C", C.c_int), ("taps", C.c_int), ("up", C.c_int), ("ss_off", C.c_int)]


class FusedArgs(C.Structure):                      # csrc/unet_kernels.hpp: struct FusedArgs (sizeof checked below)
    _fields_ = [("seg", Seg * 4), ("nseg", C.c_int), ("ss", C.c_uint64), ("ssC", C.c_int), ("silu", C.c_int),
                ("Wgt", C.c_uint64), ("Ktot", C.c_int), ("bias", C.c_uint64), ("temb", C.c_uint64),
                ("temb_bstride", C.c_int), ("temb_off", C.c_int), ("resid", C.c_uint64), ("out", C.c_uint64),
                ("out_nchw32", C.c_int), ("nco", C.c_int), ("stats", C.c_uint64), ("B", C.c_int), ("H", C.c_int),
                ("W", C.c_int), ("Cout", C.c_int), ("zeros", C.c_uint64), ("gn_p1", C.c_uint64), ("gn_p2", C.c_uint64),
                ("gn_gamma", C.c_uint64), ("gn_beta", C.c_uint64), ("gn_ns1", C.c_int), ("gn_ns2", C.c_int),
                ("gn_C1", C.c_int), ("gn_HW", C.c_int), ("gn_eps", C.c_float)]


assert C.sizeof(FusedArgs) == 272


def dev(addr, dtype, n):
    """numpy view of n elements of the stand-in's device memory"""
    return np.ctypeslib.as_array(C.cast(C.c_void_p(addr), C.POINTER(C.c_uint8)), shape=(n * np.dtype(dtype).itemsize,)).view(dtype)


def silu(x):
    return x / (1.0 + np.exp(-x))


def conv3x3(x, w):
    """x [H, W, Cin] fp32, w [Cout, Cin, 3, 3] -> [H, W, Cout], zero padding 1"""
    Hh, Ww, Ci = x.shape
    xp = np.zeros((Hh + 2, Ww + 2, Ci), np.float32)
    xp[1:-1, 1:-1] = x
    out = np.zeros((Hh, Ww, w.shape[0]), np.float32)
    for ky in range(3):
        for kx in range(3):
            out += (xp[ky:ky + Hh, kx:kx + Ww].reshape(-1, Ci) @ np.ascontiguousarray(w[:, :, ky, kx].T)).reshape(Hh, Ww, -1)
    return out


def up2(x):
    return x.repeat(2, axis=0).repeat(2, axis=1)


def group_norm(x, gamma, beta, eps, groups=32):
    """x [H, W, C] fp32 -> GroupNorm(32) over (H, W, C / 32), fp64 statistics"""
    Hh, Ww, Cc = x.shape
    g = x.reshape(Hh * Ww, groups, Cc // groups).astype(np.float64)
    mean = g.mean(axis=(0, 2), keepdims=True)
    var = g.var(axis=(0, 2), keepdims=True)
    y = ((g - mean) / np.sqrt(var + eps)).reshape(Hh, Ww, Cc)
    return (y * gamma + beta).astype(np.float32)


def decode_weights(a, rows, read16=None):
    """packed upload -> (W9 [Cout][sum of 3x3 channels][9], W1 [Cout][sum of 1x1 channels]) as the K loop consumes them"""
    nsteps = sum(a.seg[i].taps * (a.seg[i].C // 32) for i in range(a.nseg))
    ntn = (a.Cout + rows - 1) // rows
    read16 = read16 or (lambda addr, n: dev(addr, np.float16, n).astype(np.float32))
    wp = read16(a.Wgt, ntn * nsteps * rows * 32).reshape(ntn, nsteps, rows, 4, 8)
    r = np.arange(rows)
    C9 = sum(a.seg[i].C for i in range(a.nseg) if a.seg[i].taps == 9)
    C1 = sum(a.seg[i].C for i in range(a.nseg) if a.seg[i].taps == 1)
    W9 = np.zeros((ntn * rows, C9, 9), np.float32)
    W1 = np.zeros((ntn * rows, C1), np.float32)
    step = 0
    c9 = c1 = 0
    for taps in (9, 1):                                        # 3x3 segments first, in FusedArgs order
        for i in range(a.nseg):
            if a.seg[i].taps != taps:
                continue
            for ci in range(a.seg[i].C // 32):
                for t in range(taps):
                    for j in range(4):                         # physical slot j of row r holds channel group j ^ ((r >> 2) & 3)
                        s = j ^ ((r >> 2) & 3)
                        for nt in range(ntn):
                            for e in range(8):
                                if taps == 9:
                                    W9[nt * rows + r, c9 + ci * 32 + s * 8 + e, t] = wp[nt, step, r, j, e]
                                else:
                                    W1[nt * rows + r, c1 + ci * 32 + s * 8 + e] = wp[nt, step, r, j, e]
                    step += 1
            if taps == 9:
                c9 += a.seg[i].C
            else:
                c1 += a.seg[i].C
    return W9[:a.Cout], W1[:a.Cout]


def main():
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
    lib = _lib.load()
    d = drive.Dev()
    rs = np.random.RandomState(11)
    h = C.c_void_p()
    cfg = drive.unet_cfg(3, 6, 64, *drive.RES64, drive.F16, 1)
    _lib.check(lib.bndm_unet_create(C.byref(h), C.byref(cfg)), "create")
    sd = {}
    name, numel = C.create_string_buffer(200), C.c_int64()
    for i in range(lib.bndm_unet_num_params(h)):
        _lib.check(lib.bndm_unet_param_info(h, i, name, 200, C.byref(numel)), "param_info")
        w = (rs.standard_normal(numel.value) * 0.05).astype(np.float32)
        if name.value.endswith(b"weight") and b"norm" in name.value.split(b".")[-2]:
            w += 1.0
        sd[name.value.decode()] = w
        _lib.check(lib.bndm_unet_load_param(h, name.value, w.ctypes.data_as(C.c_void_p), numel.value), "load_param")
    _lib.check(lib.bndm_unet_finalize(h), "finalize")
    drive.mark("forward")
    drive.forward(lib, d, h, 1, 3, 6, 64)
    d.flush()
    lines = open(os.environ["HIPMOCK_TRACE"]).read().splitlines()
    launches = [H.parse_launch(x) for x in dict(H.stages(lines))["forward"] if x.startswith("launch ")]
    t32 = [x for x in launches if x["name"] == "conv_t32"]
    kern, lab, fl = C.create_string_buffer(128), C.create_string_buffer(256), C.c_double()
    labels = []
    for i in range(lib.bndm_unet_num_ops(h)):
        _lib.check(lib.bndm_unet_op_info(h, i, kern, 128, lab, 256, C.byref(fl)), "op_info")
        if kern.value.startswith(b"conv_t32"):
            labels.append(" ".join(lab.value.decode().split()))
    assert len(labels) == len(t32), (len(labels), len(t32))

    worst = 0.0
    for label, ln in zip(labels, t32):
        a = FusedArgs.from_buffer_copy(ln["args"][0])
        head = a.out_nchw32 != 0
        mod = label.split()[1] if not head else "head"
        Hh, Ww, B = a.H, a.W, a.B
        assert B == 1
        # ---- fill the launch's inputs ------------------------------------------------------------------
        xs = []
        for i in range(a.nseg):
            s = a.seg[i]
            hs, wsz = (Hh // 2, Ww // 2) if s.up else (Hh, Ww)
            buf = dev(s.src, np.float16, hs * wsz * s.C)
            buf[:] = rs.standard_normal(buf.size).astype(np.float16)
            xs.append(buf.astype(np.float32).reshape(hs, wsz, s.C))
        normed = a.gn_p1 != 0
        if normed:
            C1, C2 = a.gn_C1, a.ssC - a.gn_C1
            p1 = dev(a.gn_p1, np.float32, a.gn_ns1 * C1 * 2).reshape(a.gn_ns1, C1, 2)
            p1[:] = 0
            p2 = dev(a.gn_p2, np.float32, a.gn_ns2 * C2 * 2).reshape(a.gn_ns2, C2, 2) if C2 else None
            if p2 is not None:
                p2[:] = 0
            for i in range(a.nseg):
                s = a.seg[i]
                if s.ss_off < 0:
                    continue
                flat = xs[i].reshape(-1, s.C)
                assert flat.shape[0] == a.gn_HW, (label, flat.shape, a.gn_HW)
                tgt, off = (p1, s.ss_off) if s.ss_off < C1 else (p2, s.ss_off - C1)
                # spread over the slabs as the producers would (any split sums to the same totals)
                k = tgt.shape[0]
                for q in range(k):
                    part = flat[q::k]
                    tgt[q, off:off + s.C, 0] = part.sum(0)
                    tgt[q, off:off + s.C, 1] = (part.astype(np.float64) ** 2).sum(0)
        if a.temb:
            row = dev(a.temb + 4 * a.temb_off, np.float32, a.Cout)
            row[:] = rs.standard_normal(a.Cout).astype(np.float32)
        elif a.bias:
            row = dev(a.bias, np.float32, a.Cout).copy()
        else:
            row = np.zeros(a.Cout, np.float32)
        add = np.array(row, np.float32)
        resid = None
        if a.resid:
            rb = dev(a.resid, np.float16, Hh * Ww * a.Cout)
            rb[:] = rs.standard_normal(rb.size).astype(np.float16)
            resid = rb.astype(np.float32).reshape(Hh, Ww, a.Cout)

        # ---- MODEL: as the kernel arguments and the uploaded tables say --------------------------------
        if normed:
            gam = dev(a.gn_gamma, np.float32, a.ssC)
            bet = dev(a.gn_beta, np.float32, a.ssC)
            tot = np.concatenate([p1.sum(0)] + ([p2.sum(0)] if p2 is not None else []), 0).astype(np.float64)   # [ssC][2]
            Cg = a.ssC // 32
            g = tot.reshape(32, Cg, 2).sum(1)
            n = Cg * a.gn_HW
            mean = g[:, 0] / n
            var = np.maximum(g[:, 1] / n - mean * mean, 0)
            rstd = 1.0 / np.sqrt(var + a.gn_eps)
            sc = (np.repeat(rstd, Cg) * gam).astype(np.float32)
            sh = (bet - np.repeat(mean, Cg).astype(np.float32) * sc).astype(np.float32)
        parts9, parts1 = [], []
        for i in range(a.nseg):
            s = a.seg[i]
            x = xs[i]
            if s.taps == 9:
                if s.ss_off >= 0:
                    y = x * sc[s.ss_off:s.ss_off + s.C] + sh[s.ss_off:s.ss_off + s.C]
                    y = silu(y) if a.silu else y
                    x = (LOG2E * y).astype(np.float16).astype(np.float32)        # the patch holds log2(e) * silu(.), 16-bit
                parts9.append(up2(x) if s.up else x)
            else:
                assert s.ss_off < 0
                parts1.append(up2(x) if s.up else x)
        W9, W1 = decode_weights(a, 32 if head else 128)
        acc = conv3x3(np.concatenate(parts9, -1), W9.reshape(a.Cout, -1, 3, 3))
        if parts1:
            acc += (np.concatenate(parts1, -1).reshape(Hh * Ww, -1) @ W1.T).reshape(Hh, Ww, -1)
        model = acc + add
        if not head:
            model = model.astype(np.float16).astype(np.float32)
            if resid is not None:
                model = (model + resid).astype(np.float16).astype(np.float32)

        # ---- REFERENCE: the module's definition on the state dict's original tensors ------------------
        def P(k):
            return sd[k]

        cat = np.concatenate([up2(xs[i]) if a.seg[i].up else xs[i] for i in range(a.nseg) if a.seg[i].taps == 9], -1)
        Cin9 = cat.shape[-1]
        if head:
            y = silu(group_norm(cat, P("conv_norm_out.weight"), P("conv_norm_out.bias"), 1e-5))
            ref = conv3x3(y, P("conv_out.weight").reshape(a.Cout, Cin9, 3, 3)) + P("conv_out.bias")
        elif mod.endswith(".conv"):                         # Upsample2D: nearest 2x, then the conv
            assert not normed
            ref = conv3x3(cat, P(mod + ".weight").reshape(a.Cout, Cin9, 3, 3)) + P(mod + ".bias")
        elif mod.endswith(".conv1"):
            blk = mod[:-len(".conv1")]
            y = silu(group_norm(cat, P(blk + ".norm1.weight"), P(blk + ".norm1.bias"), 1e-5))
            ref = conv3x3(y, P(blk + ".conv1.weight").reshape(a.Cout, Cin9, 3, 3)) + add      # (+ the time-embedding row as given)
        else:
            blk = re.sub(r"\.conv2(\+sc)?$", "", mod)
            y = silu(group_norm(cat, P(blk + ".norm2.weight"), P(blk + ".norm2.bias"), 1e-5))
            ref = conv3x3(y, P(blk + ".conv2.weight").reshape(a.Cout, Cin9, 3, 3)) + P(blk + ".conv2.bias")
            if mod.endswith("+sc"):
                xin = np.concatenate([xs[i] for i in range(a.nseg) if a.seg[i].taps == 1], -1)
                ref += (xin.reshape(Hh * Ww, -1) @ P(blk + ".conv_shortcut.weight").reshape(a.Cout, -1).T).reshape(Hh, Ww, -1) \
                    + P(blk + ".conv_shortcut.bias")
            else:
                assert resid is not None
                ref += resid
        err = float(np.linalg.norm((model - ref).astype(np.float64)) / np.linalg.norm(ref.astype(np.float64)))
        worst = max(worst, err)
        print(f"{err:.2e}  {label}")
        assert err <= 5e-3, f"{label}: model vs module definition rel-L2 {err:.3e}"
    print(f"OK {len(t32)} conv_t32 launches, worst rel-L2 {worst:.2e}")


if __name__ == "__main__":
    main()
