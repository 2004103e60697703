"""TEST INFRASTRUCTURE: one whole UNet forward executed on the CPU THROUGH THE ENGINE'S OWN LAUNCH LIST.

Under the recording HIP runtime (tests/hipmock) the library's bndm_unet_forward leaves a trace of every launch with its
argument bytes, and its uploads sit in readable "device" memory.  This script replays that trace: each launch is handed to
a numpy model of its kernel's CONTRACT -- what the kernel reads (sources, packed weights, step lists, partial sums, rows),
what it writes (activations, GroupNorm partial sums and normalised copies, split-K slabs, the fp32 output) and the arithmetic
in between, with the 16-bit roundings where the kernel rounds -- and the models read and write the same buffers the real
kernels would.  The final fp32 output is saved; tests/test_launch_trace.py compares it with oracle/unet_oracle.py on the same
weights and inputs.

What agreement means: the launch list, its order, every buffer hand-over between launches (skip connections, concatenations,
normalised copies, partial sums, split-K slabs, the time-embedding table), every packed weight / table layout and every
per-launch parameter are right for a forward at batch 2 -- the whole HOST side of the engine.  The kernels' device code is not
involved at any point.

    LD_LIBRARY_PATH=<stand-in dir> HIPMOCK_TRACE=t.txt HIPMOCK_KERNARGS=ka.txt python tests/hipmock/exec_forward.py lib.so out_dir
"""
import ctypes as C
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from bndm_amd import _lib  # noqa: E402
from tests.hipmock import drive, harness as H  # noqa: E402
from tests.hipmock.check_conv_s import Round, TailArgs  # noqa: E402
from tests.hipmock.check_conv_t32 import LOG2E, FusedArgs, conv3x3, decode_weights, dev, silu, up2  # noqa: E402

f16 = np.float16
# candidate layout (tools/experiments/round4_pairstats_sumsfirst_th32.patch): GroupNorm partial sums per channel PAIR,
# [B][slabs][C / 2][2] = (sum, sum of squares) over the slab's pixels and both channels of the pair
PAIR = os.environ.get("EXEC_PAIRSTATS") == "1"


STAT_SHAPE = {}                                            # partial-sum buffer -> (slabs per sample, channels) as its producer wrote it


def stat_write(st_addr, v, B, nslab):
    """v [B, nslab, pixels, C] stored values -> the partial-sum buffer in the layout in force"""
    Cc = v.shape[-1]
    STAT_SHAPE[st_addr] = (nslab, Cc)
    if PAIR:
        st = dev(st_addr, np.float32, B * nslab * Cc).reshape(B, nslab, Cc // 2, 2)
        p = v.reshape(B, nslab, -1, Cc // 2, 2)
        st[..., 0] = p.sum((2, 4))
        st[..., 1] = (p.astype(np.float64) ** 2).sum((2, 4))
    else:
        st = dev(st_addr, np.float32, B * nslab * Cc * 2).reshape(B, nslab, Cc, 2)
        st[..., 0] = v.sum(2)
        st[..., 1] = (v.astype(np.float64) ** 2).sum(2)


def stat_write_into(dst, v):
    """dst [B, units, 2] <- sums of v [B, pixels, C] (units = channels or channel pairs)"""
    if PAIR:
        p = v.reshape(v.shape[0], v.shape[1], -1, 2)
        dst[..., 0] = p.sum((1, 3))
        dst[..., 1] = (p.astype(np.float64) ** 2).sum((1, 3))
    else:
        dst[..., 0] = v.sum(1)
        dst[..., 1] = (v.astype(np.float64) ** 2).sum(1)


def stat_totals(addr, B, ns, Cc):
    """-> [B, units, 2] sums over the slabs; units = channels, or channel pairs in the candidate layout"""
    assert STAT_SHAPE.get(addr) == (ns, Cc), f"consumer reads {ns} slabs x {Cc} channels at {addr:#x}, producer wrote {STAT_SHAPE.get(addr)}"
    u = Cc // 2 if PAIR else Cc
    return dev(addr, np.float32, B * ns * u * 2).reshape(B, ns, u, 2).sum(1).astype(np.float64)


BF16 = False                                               # set by main() from the case: 16-bit storage type of the handle


def to16(x):
    """fp32 array -> the handle's 16-bit storage (uint16 bit patterns), round to nearest even"""
    x = np.ascontiguousarray(x, np.float32)
    if not BF16:
        return x.astype(f16).view(np.uint16)
    u = x.view(np.uint32)
    return ((u + (0x7FFF + ((u >> 16) & 1))) >> 16).astype(np.uint16)


def from16(u):
    u = np.asarray(u, np.uint16)
    return (u.astype(np.uint32) << 16).view(np.float32) if BF16 else u.view(f16).astype(np.float32)


def rd16(addr, n):
    return from16(dev(addr, np.uint16, n))


def wr16(addr, x):
    dev(addr, np.uint16, x.size)[:] = to16(x).ravel()


def r16(x):
    return from16(to16(x)).reshape(x.shape)


class CSeg(C.Structure):
    _fields_ = [("src", C.c_uint64), ("C", C.c_int), ("taps", C.c_int), ("up", C.c_int), ("pad", C.c_int)]


class ConvArgs(C.Structure):                       # csrc/unet_kernels.hpp: struct ConvArgs
    _fields_ = [("seg", CSeg * 4), ("nseg", C.c_int), ("Wgt", C.c_uint64), ("bias", C.c_uint64), ("temb", C.c_uint64),
                ("temb_bstride", C.c_int), ("temb_off", C.c_int), ("resid", C.c_uint64), ("out", C.c_uint64), ("B", C.c_int),
                ("H", C.c_int), ("W", C.c_int), ("stride", C.c_int), ("Cout", C.c_int), ("Ktot", C.c_int), ("splitk", C.c_int),
                ("zeros", C.c_uint64), ("steps", C.c_uint64), ("counters", C.c_uint64), ("wtiled", C.c_int), ("wmajor", C.c_int)]


class SlabSrc(C.Structure):                        # struct GnSlabSrc
    _fields_ = [("part", C.c_uint64), ("splitk", C.c_int), ("bias", C.c_uint64), ("temb", C.c_uint64), ("temb_bstride", C.c_int),
                ("temb_off", C.c_int), ("resid", C.c_uint64), ("raw_out", C.c_uint64)]


assert C.sizeof(ConvArgs) == 216


def u64(b):
    return int.from_bytes(b, "little")


def i32(b):
    return int.from_bytes(b, "little", signed=True)


def f32(b):
    return float(np.frombuffer(b, np.float32)[0])


# ------------------------------------------------------------------------------------------------------------------
def k_temb_mlp(L):
    t, C0, D, W1t, b1, W2t, b2, act = [L["args"][i] for i in range(8)]
    t, C0, D, W1t, b1, W2t, b2, act = u64(t), i32(C0), i32(D), u64(W1t), u64(b1), u64(W2t), u64(b2), u64(act)
    B = int(L["g"].split(",")[1])
    tv = dev(t, np.float32, B).astype(np.float64)
    half = C0 // 2
    f = np.exp(-9.210340371976184 * np.arange(half) / half)
    ang = tv[:, None] * f[None]
    emb = np.concatenate([np.cos(ang), np.sin(ang)], 1).astype(np.float32)            # flip_sin_to_cos
    h1 = silu(emb @ dev(W1t, np.float32, C0 * D).reshape(C0, D) + dev(b1, np.float32, D))
    v = h1 @ dev(W2t, np.float32, D * D).reshape(D, D) + dev(b2, np.float32, D)
    wr16(act, silu(v))


def igemm_weights(a):
    rows = -(-a.Cout // 128) * 128 if a.wtiled else None
    if a.wtiled:
        ks = a.Ktot // 64
        wt = rd16(a.Wgt, rows * a.Ktot).reshape(rows // 128, ks, 128, 8, 8)
        wp = np.zeros((rows, a.Ktot), np.float32)
        r = np.arange(128)
        for j in range(8):
            g = j ^ ((r >> 1) & 7)                                                    # slot j of row r holds k-group g
            for nt in range(rows // 128):
                for s in range(ks):
                    for rr in range(128):
                        wp[nt * 128 + rr, s * 64 + g[rr] * 8:s * 64 + g[rr] * 8 + 8] = wt[nt, s, rr, j]
        return wp[:a.Cout]
    # plain [Cout_pad][Ktot]: the row padding is unknown here, but only the first Cout rows matter
    return rd16(a.Wgt, a.Cout * a.Ktot).reshape(a.Cout, a.Ktot)


def k_igemm(L):
    a = ConvArgs.from_buffer_copy(L["args"][0])
    epi = int(re.search(r"conv_igemmI\w+?_?Li\d+ELi\d+ELi\d+ELi\d+ELi(\d+)E", L["sym"]).group(1))
    Wm = igemm_weights(a)                                                             # [Cout][Ktot], k: segment -> tap -> channel
    B, Hh, Ww = a.B, a.H, a.W
    M = B * Hh * Ww
    acc = np.zeros((M, a.Cout), np.float32)
    koff = 0
    for i in range(a.nseg):
        s = a.seg[i]
        if a.stride == 2:
            hs, ws = 2 * Hh, 2 * Ww
        elif s.up:
            hs, ws = Hh // 2, Ww // 2
        else:
            hs, ws = Hh, Ww
        x = rd16(s.src, B * hs * ws * s.C).reshape(B, hs, ws, s.C)
        Wseg = Wm[:, koff:koff + s.taps * s.C].reshape(a.Cout, s.taps, s.C)
        for b in range(B):
            xb = up2(x[b]) if s.up else x[b]
            if s.taps == 9:
                w4 = Wseg.transpose(0, 2, 1).reshape(a.Cout, s.C, 3, 3)
                full = conv3x3(xb, w4)
                if a.stride == 2:
                    full = full[0::2, 0::2]
                acc[b * Hh * Ww:(b + 1) * Hh * Ww] += full.reshape(Hh * Ww, -1)
            else:
                assert a.stride == 1
                acc[b * Hh * Ww:(b + 1) * Hh * Ww] += xb.reshape(Hh * Ww, s.C) @ Wseg[:, 0].T
        koff += s.taps * s.C
    if epi == 2:                                           # EPI_NCHW32: the network head on the fallback path, fp32 NCHW + bias
        o = acc + (dev(a.bias, np.float32, a.Cout) if a.bias else 0.0)
        dev(a.out, np.float32, M * a.Cout)[:] = o.reshape(B, Hh * Ww, a.Cout).transpose(0, 2, 1).ravel()
        return
    if a.splitk > 1:                                       # fp32 slabs [splitk][M][Cout], no bias: the whole sum in slab 0
        slabs = dev(a.out, np.float32, a.splitk * M * a.Cout).reshape(a.splitk, M, a.Cout)
        slabs[:] = 0
        slabs[0] = acc
        STAT_SHAPE[a.out] = ("splitk", a.splitk, M, a.Cout)
        return
    if a.bias:
        acc += dev(a.bias, np.float32, a.Cout)
    if a.temb:
        for b in range(B):
            acc[b * Hh * Ww:(b + 1) * Hh * Ww] += dev(a.temb + 4 * (b * a.temb_bstride + a.temb_off), np.float32, a.Cout)
    if epi == 1:                                           # EPI_F32_ROWS
        assert not a.resid
        dev(a.out, np.float32, M * a.Cout)[:] = acc.ravel()
        return
    assert epi == 0
    if a.resid:
        acc += rd16(a.resid, M * a.Cout).reshape(M, a.Cout)
    wr16(a.out, acc)


def k_splitk_reduce(L):
    part, splitk = u64(L["args"][0]), i32(L["args"][1])
    a = ConvArgs.from_buffer_copy(L["args"][2])
    logHW = i32(L["args"][3])
    M = a.B << logHW
    assert STAT_SHAPE.get(part) == ("splitk", splitk, M, a.Cout), (STAT_SHAPE.get(part), splitk, M, a.Cout)
    s = dev(part, np.float32, splitk * M * a.Cout).reshape(splitk, M, a.Cout).sum(0)
    if a.bias:
        s = s + dev(a.bias, np.float32, a.Cout)
    if a.temb:
        for b in range(a.B):
            s[b << logHW:(b + 1) << logHW] += dev(a.temb + 4 * (b * a.temb_bstride + a.temb_off), np.float32, a.Cout)
    if a.resid:
        s = s + rd16(a.resid, M * a.Cout).reshape(M, a.Cout)
    wr16(a.out, s)


def k_conv_in(L):
    A = L["args"]
    x, Cx, extra, Ce, W16, bias, out, stats, B, logH, logW, C0, KP = (u64(A[0]), i32(A[1]), u64(A[2]), i32(A[3]), u64(A[4]), u64(A[5]),
                                                                         u64(A[6]), u64(A[7]), i32(A[8]), i32(A[9]), i32(A[10]), i32(A[11]), i32(A[12]))
    Hh, Ww = 1 << logH, 1 << logW
    xin = dev(x, np.float32, B * Cx * Hh * Ww).reshape(B, Cx, Hh, Ww)
    if Ce:
        xin = np.concatenate([xin, dev(extra, np.float32, B * Ce * Hh * Ww).reshape(B, Ce, Hh, Ww)], 1)
    Cin = Cx + Ce
    W = rd16(W16, C0 * KP).reshape(C0, KP)[:, :9 * Cin].reshape(C0, Cin, 3, 3)   # k = ci * 9 + ky * 3 + kx
    o = np.stack([r16((conv3x3(r16(xin[b].transpose(1, 2, 0)), W) + dev(bias, np.float32, C0)).reshape(Hh * Ww, C0)) for b in range(B)])
    wr16(out, o)
    if stats:
        ns = (Hh * Ww) >> 7
        stat_write(stats, o.reshape(B, ns, 128, C0), B, ns)


def k_gn_stats(L):
    A = L["args"]
    x1, C1, x2, C2, HW, partial, nslab = u64(A[0]), i32(A[1]), u64(A[2]), i32(A[3]), i32(A[4]), u64(A[5]), i32(A[6])
    B = int(L["g"].split(",")[1])
    x = rd16(x1, B * HW * C1).reshape(B, HW, C1)
    if C2:
        x = np.concatenate([x, rd16(x2, B * HW * C2).reshape(B, HW, C2)], -1)
    Cc = C1 + C2
    stat_write(partial, x.reshape(B, nslab, HW // nslab, Cc), B, nslab)


def k_gn_finalize2(L):
    A = L["args"]
    p1, ns1, C1, p2, ns2, C2, HW, groups, eps, gamma, beta, ss = (u64(A[0]), i32(A[1]), i32(A[2]), u64(A[3]), i32(A[4]), i32(A[5]), i32(A[6]),
                                                                  i32(A[7]), f32(A[8]), u64(A[9]), u64(A[10]), u64(A[11]))
    B = int(L["g"].split(",")[1])
    Cc = C1 + C2
    tot = stat_totals(p1, B, ns1, C1)
    if C2:
        tot = np.concatenate([tot, stat_totals(p2, B, ns2, C2)], 1)
    Cg = Cc // groups
    g = tot.reshape(B, groups, -1, 2).sum(2)
    n = Cg * HW
    mean = g[..., 0] / n
    var = np.maximum(g[..., 1] / n - mean * mean, 0)
    sc = (np.repeat(1.0 / np.sqrt(var + eps), Cg, axis=1) * dev(gamma, np.float32, Cc)).astype(np.float32)
    sh = (dev(beta, np.float32, Cc) - np.repeat(mean, Cg, axis=1).astype(np.float32) * sc).astype(np.float32)
    out = dev(ss, np.float32, B * 2 * Cc).reshape(B, 2, Cc)
    out[:, 0], out[:, 1] = sc, sh


def k_gn_small(L):
    A = L["args"]
    x1, C1, x2, C2, HW, groups, eps, gamma, beta, act, out = (u64(A[0]), i32(A[1]), u64(A[2]), i32(A[3]), i32(A[4]), i32(A[5]), f32(A[6]),
                                                               u64(A[7]), u64(A[8]), i32(A[9]), u64(A[10]))
    sl = SlabSrc.from_buffer_copy(A[11])
    B = int(L["g"].split(",")[1])                         # grid (8, B)
    M = B * HW
    if sl.part:                                            # x1 arrives as split-K slabs (+ bias + time embedding + residual)
        assert STAT_SHAPE.get(sl.part) == ("splitk", sl.splitk, M, C1), (STAT_SHAPE.get(sl.part), sl.splitk, M, C1)
        a = dev(sl.part, np.float32, sl.splitk * M * C1).reshape(sl.splitk, M, C1).sum(0)
        if sl.bias:
            a = a + dev(sl.bias, np.float32, C1)
        if sl.temb:
            for b in range(B):
                a[b * HW:(b + 1) * HW] += dev(sl.temb + 4 * (b * sl.temb_bstride + sl.temb_off), np.float32, C1)
        if sl.resid:
            a = a + rd16(sl.resid, M * C1).reshape(M, C1)
        a = r16(a)
        if sl.raw_out:
            wr16(sl.raw_out, a)
    else:
        a = rd16(x1, M * C1).reshape(M, C1)
    if C2:
        a = np.concatenate([a, rd16(x2, M * C2).reshape(M, C2)], -1)
    Cc = C1 + C2
    gs = Cc // groups
    g = a.reshape(B, HW, groups, gs).astype(np.float64)
    mean = g.mean(axis=(1, 3), keepdims=True)
    var = g.var(axis=(1, 3), keepdims=True)
    y = ((g - mean) / np.sqrt(var + eps)).reshape(M, Cc).astype(np.float32) * dev(gamma, np.float32, Cc) + dev(beta, np.float32, Cc)
    wr16(out, (silu(y) if act else y))


def k_conv_t32(L):
    a = FusedArgs.from_buffer_copy(L["args"][0])
    TH = int(re.search(r"conv_t32I\w+?_?Li(\d+)E", L["sym"]).group(1))
    head = a.out_nchw32 != 0
    B, Hh, Ww = a.B, a.H, a.W
    W9, W1 = decode_weights(a, 32 if head else 128, rd16)
    xs = []
    for i in range(a.nseg):
        s = a.seg[i]
        hs, ws = (Hh // 2, Ww // 2) if s.up else (Hh, Ww)
        xs.append(rd16(s.src, B * hs * ws * s.C).reshape(B, hs, ws, s.C))
    normed = a.gn_p1 != 0
    table = dev(a.ss, np.float32, B * 2 * a.ssC).reshape(B, 2, a.ssC) if (a.ss and not normed) else None   # gn_finalize2's table
    if normed:
        C1, C2 = a.gn_C1, a.ssC - a.gn_C1
        tot = stat_totals(a.gn_p1, B, a.gn_ns1, C1)
        if C2:
            tot = np.concatenate([tot, stat_totals(a.gn_p2, B, a.gn_ns2, C2)], 1)
        gam, bet = dev(a.gn_gamma, np.float32, a.ssC), dev(a.gn_beta, np.float32, a.ssC)
        Cg = a.ssC // 32
    outs = []
    for b in range(B):
        if normed:
            g = tot[b].reshape(32, -1, 2).sum(1)
            n = Cg * a.gn_HW
            mean = g[:, 0] / n
            var = np.maximum(g[:, 1] / n - mean * mean, 0)
            sc = (np.repeat(1.0 / np.sqrt(var + a.gn_eps), Cg) * gam).astype(np.float32)
            sh = (bet - np.repeat(mean, Cg).astype(np.float32) * sc).astype(np.float32)
        elif table is not None:
            sc, sh = table[b, 0], table[b, 1]
        p9, p1 = [], []
        for i in range(a.nseg):
            s, x = a.seg[i], xs[i][b]
            if s.taps == 9 and s.ss_off >= 0:
                y = x * sc[s.ss_off:s.ss_off + s.C] + sh[s.ss_off:s.ss_off + s.C]
                x = r16(LOG2E * (silu(y) if a.silu else y))
            (p9 if s.taps == 9 else p1).append(up2(x) if s.up else x)
        acc = conv3x3(np.concatenate(p9, -1), W9.reshape(a.Cout, -1, 3, 3))
        if p1:
            acc += (np.concatenate(p1, -1).reshape(Hh * Ww, -1) @ W1.T).reshape(Hh, Ww, -1)
        if a.temb:
            acc += dev(a.temb + 4 * (b * a.temb_bstride + a.temb_off), np.float32, a.Cout)
        elif a.bias:
            acc += dev(a.bias, np.float32, a.Cout)
        outs.append(acc)
    y = np.stack(outs)
    if head:
        dev(a.out, np.float32, B * a.Cout * Hh * Ww)[:] = y.transpose(0, 3, 1, 2).ravel()
        return
    y = r16(y)
    if a.resid:
        y = r16(y + rd16(a.resid, B * Hh * Ww * a.Cout).reshape(B, Hh, Ww, a.Cout))
    wr16(a.out, y)
    if a.stats:
        tps = (Hh // TH) * (Ww // 16)
        STAT_SHAPE[a.stats] = (tps, a.Cout)
        u = a.Cout // 2 if PAIR else a.Cout                 # (the consumer adds the tiles up: the whole sample in tile 0)
        dev(a.stats, np.float32, B * tps * u * 2)[:] = 0
        tmp = dev(a.stats, np.float32, B * tps * u * 2).reshape(B, tps, u, 2)
        stat_write_into(tmp[:, 0], y.reshape(B, Hh * Ww, a.Cout))


class HeadArgs(C.Structure):                       # candidate (tools/experiments/head_conv_kernel.patch): csrc/unet_kernels.hpp struct HeadArgs
    _fields_ = [("x", C.c_uint64), ("gn_p", C.c_uint64), ("gn_gamma", C.c_uint64), ("gn_beta", C.c_uint64), ("gn_ns", C.c_int),
                ("gn_HW", C.c_int), ("gn_eps", C.c_float), ("Wgt", C.c_uint64), ("bias", C.c_uint64), ("out", C.c_uint64), ("B", C.c_int),
                ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("Cout", C.c_int), ("ex", C.c_uint64), ("eda", C.c_float), ("edg", C.c_float),
                ("eC", C.c_int)]


def k_head_conv(L):
    """contract of head_conv: GroupNorm(32) from the producer's partial sums, log2(e) * SiLU rounded to 16 bits, 3x3 convolution with the
    packed weights [9 C / 32 K-steps (chunk-major, tap-minor)][8 rows][32 k] (ln 2 folded in), + bias -> fp32 NCHW, or the Euler update"""
    a = HeadArgs.from_buffer_copy(L["args"][0])
    B, Hh, Ww, Cc, Co = a.B, a.H, a.W, a.C, a.Cout
    x = rd16(a.x, B * Hh * Ww * Cc).reshape(B, Hh, Ww, Cc)
    tot = stat_totals(a.gn_p, B, a.gn_ns, Cc)
    gam, bet = dev(a.gn_gamma, np.float32, Cc), dev(a.gn_beta, np.float32, Cc)
    wp = rd16(a.Wgt, 9 * (Cc // 32) * 8 * 32).reshape(Cc // 32, 9, 8, 32)
    Wd = wp.transpose(2, 0, 3, 1).reshape(8, Cc, 3, 3)[:Co]                      # [co][ci = 32 chunk + k][3][3]
    Cg = Cc // 32
    outs = []
    for b in range(B):
        g = tot[b].reshape(32, -1, 2).sum(1)
        n = Cg * a.gn_HW
        mean = g[:, 0] / n
        var = np.maximum(g[:, 1] / n - mean * mean, 0)
        sc = (np.repeat(1.0 / np.sqrt(var + a.gn_eps), Cg) * gam).astype(np.float32)
        sh = (bet - np.repeat(mean, Cg).astype(np.float32) * sc).astype(np.float32)
        act = r16(LOG2E * silu(x[b] * sc + sh))
        outs.append(conv3x3(act, Wd) + (dev(a.bias, np.float32, Co) if a.bias else 0))
    dd = np.stack(outs).transpose(0, 3, 1, 2)                                     # [B][Co][H][W]
    if a.ex:
        xv = dev(a.ex, np.float32, B * a.eC * Hh * Ww).reshape(B, a.eC, Hh, Ww)
        r = xv + np.float32(a.eda) * dd[:, :a.eC]
        if Co == 2 * a.eC:
            r = r + np.float32(a.edg) * dd[:, a.eC:]
        xv[:] = r
        return
    dev(a.out, np.float32, B * Co * Hh * Ww)[:] = dd.ravel()


def k_conv_s(L):
    a = TailArgs.from_buffer_copy(L["args"][0])
    TM, NB, D = map(int, re.search(r"conv_sID[^_]*_?Li(\d+)ELi(\d+)ELi(\d+)E", L["sym"]).groups())
    HW, Wd = 1 << a.hwlog, 1 << a.wlog
    Hd, M = HW // Wd, a.B * HW
    TN, NL = NB * 32, 2 * NB
    attn = a.epi == 1
    ncol = a.ntn * TN
    rt = [Round.from_buffer_copy(bytes(dev(a.rounds + 32 * r, np.uint8, 32))) for r in range(a.nrounds)]

    def rows_of(r):
        Cs = r.row_bytes // 2
        rows_src = M if r.mode == 0 else (M >> 2 if r.mode == 1 else M << 2)
        x = rd16(r.src, rows_src * Cs).reshape(rows_src, Cs)
        m = np.arange(M)
        b, pix = m >> a.hwlog, m & (HW - 1)
        y, xx = pix >> a.wlog, pix & (Wd - 1)
        if r.mode == 0:
            return x
        if r.mode == 1:
            return x[(b << (a.hwlog - 2)) + ((y >> 1) << (a.wlog - 1)) + (xx >> 1)]
        py, px = r.phase >> 1, r.phase & 1
        return x[(((b * Hd + y) * 2 + py) << (a.wlog + 1)) + 2 * xx + px]

    def shifted(A, ts):
        dy, dx = ts // 3 - 1, ts % 3 - 1
        g = A.reshape(a.B, Hd, Wd, -1)
        out = np.zeros_like(g)
        out[:, max(0, -dy):Hd - max(0, dy), max(0, -dx):Wd - max(0, dx)] = g[:, max(0, dy):Hd - max(0, -dy), max(0, dx):Wd - max(0, -dx)]
        return out.reshape(M, -1)

    desc = dev(a.desc, np.uint32, 8 * a.maxsteps).reshape(8, a.maxsteps)
    Wdense = {}
    for w in range(8):
        rnd = 0
        for s in range(a.maxsteps):
            e = int(desc[w, s])
            if not (e & 0x2000):
                ts, j = e & 15, (e >> 4) & 7
                Wt = Wdense.setdefault((rnd, ts), np.zeros((ncol, 256), np.float32))
                for nt in range(a.ntn):
                    frag = rd16(a.wgt + nt * a.tile_bytes + w * a.wave_bytes + s * NL * 1024, NL * 512).reshape(2, NB, 64, 8)
                    for ks in range(2):
                        for nb in range(NB):
                            for half in range(2):
                                c = 32 * j + 16 * ks + 8 * half
                                Wt[nt * TN + nb * 32:nt * TN + nb * 32 + 32, c:c + 8] += frag[ks, nb, 32 * half:32 * half + 32]
            if e & 0x100:
                rnd += 1
    acc = np.zeros((M, ncol), np.float32)
    cache = {}
    for (rnd, ts), Wt in Wdense.items():
        r = rt[rnd]
        if rnd not in cache:
            cache[rnd] = rows_of(r)[:, r.cbyte // 2:r.cbyte // 2 + 32 * r.nsub]
        acc += shifted(cache[rnd], ts) @ Wt[:, :32 * r.nsub].T
    col = np.arange(ncol)
    cb = (col % TN >> 5) * a.Cout + (col // TN) * 32 + (col % TN & 31) if attn else col
    nchan = 3 * a.Cout if attn else a.Cout
    v = acc
    if a.bias:
        v = v + dev(a.bias, np.float32, nchan)[cb]
    if a.temb:
        for b in range(a.B):
            v[b * HW:(b + 1) * HW] += dev(a.temb + 4 * (b * a.temb_bstride + a.temb_off), np.float32, nchan)[cb]
    if attn:
        qkv = v.reshape(M, a.ntn, 3, 4, 8)
        q = qkv[:, :, 0].reshape(a.B, HW, a.ntn * 4, 8) * 0.35355339059327373
        k = qkv[:, :, 1].reshape(a.B, HW, a.ntn * 4, 8)
        vv = qkv[:, :, 2].reshape(a.B, HW, a.ntn * 4, 8)
        sc = np.einsum("bthe,bshe->bhts", q, k)
        p = np.exp(sc - sc.max(-1, keepdims=True))
        p /= p.sum(-1, keepdims=True)
        wr16(a.attn_out, np.einsum("bhts,bshe->bthe", p, vv).reshape(M, a.Cout))
        return
    v = v[:, :a.Cout]
    if a.resid:
        v = v + rd16(a.resid, M * a.Cout).reshape(M, a.Cout)
    if a.raw_out:
        wr16(a.raw_out, v)
    for qi in range(a.nreq):
        rq = a.req[qi]
        g = v.reshape(a.B, HW, a.Cout // rq.gs, rq.gs).astype(np.float64)
        mean = g.mean(axis=(1, 3), keepdims=True)
        var = ((g - mean) ** 2).mean(axis=(1, 3), keepdims=True)
        y = ((g - mean) / np.sqrt(var + a.eps)).reshape(M, a.Cout).astype(np.float32) * dev(rq.gamma, np.float32, a.Cout) + dev(rq.beta, np.float32, a.Cout)
        wr16(rq.out, (silu(y) if rq.silu else y))


def k_iadb_step(L):
    A = L["args"]
    x, dd, da, dg, Cc, Cout, HW4, total4 = u64(A[0]), u64(A[1]), f32(A[2]), f32(A[3]), i32(A[4]), i32(A[5]), i32(A[6]), u64(A[7])
    HW = HW4 * 4
    B = total4 // (Cc * HW4)
    xv = dev(x, np.float32, B * Cc * HW).reshape(B, Cc, HW)
    dv = dev(dd, np.float32, B * Cout * HW).reshape(B, Cout, HW)
    r = xv + np.float32(da) * dv[:, :Cc]                       # products and sums rounded separately, as steps.hip does
    if Cout == 2 * Cc:
        r = r + np.float32(dg) * dv[:, Cc:]
    xv[:] = r


def k_ddim_step(L):
    A = L["args"]
    x, eps, sat, s1at, sap, s1ap, clip, n = u64(A[0]), u64(A[1]), f32(A[2]), f32(A[3]), f32(A[4]), f32(A[5]), f32(A[6]), u64(A[7])
    xv, e = dev(x, np.float32, n), dev(eps, np.float32, n)
    x0 = (xv - np.float32(s1at) * e) / np.float32(sat)
    if clip > 0:
        x0 = np.clip(x0, -clip, clip)
    xv[:] = np.float32(sap) * x0 + np.float32(s1ap) * e


def k_pointwise_f32(L):
    A = L["args"]
    z, w, bias, out, Cin, Cout, HW, total = u64(A[0]), u64(A[1]), u64(A[2]), u64(A[3]), i32(A[4]), i32(A[5]), i32(A[6]), u64(A[7])
    B = total // (Cout * HW)
    zz = dev(z, np.float32, B * Cin * HW).reshape(B, Cin, HW)
    o = np.einsum("mc,bcp->bmp", dev(w, np.float32, Cout * Cin).reshape(Cout, Cin), zz) + dev(bias, np.float32, Cout)[None, :, None]
    dev(out, np.float32, B * Cout * HW)[:] = o.astype(np.float32).ravel()


def k_gn_apply(L):
    A = L["args"]
    x1, C1, x2, C2, ss, HW, total_chunks, act, out = u64(A[0]), i32(A[1]), u64(A[2]), i32(A[3]), u64(A[4]), i32(A[5]), u64(A[6]), i32(A[7]), u64(A[8])
    Cc = C1 + C2
    M = total_chunks * 8 // Cc
    B = M // HW
    x = rd16(x1, M * C1).reshape(M, C1)
    if C2:
        x = np.concatenate([x, rd16(x2, M * C2).reshape(M, C2)], -1)
    t = dev(ss, np.float32, B * 2 * Cc).reshape(B, 2, Cc)
    y = x.reshape(B, HW, Cc) * t[:, None, 0] + t[:, None, 1]
    wr16(out, (silu(y) if act else y))


def k_softmax_rows(L):
    A = L["args"]
    sp, rows, n, scale = u64(A[0]), i32(A[1]), i32(A[2]), f32(A[3])
    v = rd16(sp, rows * n).reshape(rows, n) * np.float32(scale)
    p = np.exp(v - v.max(-1, keepdims=True))
    wr16(sp, p / p.sum(-1, keepdims=True))


def k_attention(L):
    A = L["args"]
    qkv, out, B, T, Cc = u64(A[0]), u64(A[1]), i32(A[2]), i32(A[3]), i32(A[4])
    m = rd16(qkv, B * T * 3 * Cc).reshape(B, T, 3, Cc // 8, 8)
    q, k, v = m[:, :, 0] * 0.35355339059327373, m[:, :, 1], m[:, :, 2]
    sc = np.einsum("bthe,bshe->bhts", q, k)
    p = np.exp(sc - sc.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    wr16(out, np.einsum("bhts,bshe->bthe", p, v))


# ---- the fp32-compute verification mode (csrc/unet_f32.hip): plain fp32 NCHW kernels --------------------------------------
def k_conv_f32(L):
    A = L["args"]
    KS = int(re.search(r"conv_f32_kernelILi(\d+)E", L["sym"]).group(1))
    xin, w, bias, addbc, resid, out = (u64(A[i]) for i in range(6))
    Cin, Hi, Wi, Cout, Ho, Wo, stride, up = (i32(A[i]) for i in range(6, 14))
    B = int(L["g"].split(",")[2])
    x = dev(xin, np.float32, B * Cin * Hi * Wi).reshape(B, Cin, Hi, Wi)
    W = dev(w, np.float32, Cout * Cin * KS * KS).reshape(Cout, Cin, KS, KS)
    o = dev(out, np.float32, B * Cout * Ho * Wo).reshape(B, Cout, Ho * Wo)
    for b in range(B):
        xb = x[b].transpose(1, 2, 0)
        xb = up2(xb) if up else xb
        if KS == 3:
            y = conv3x3(xb, W)
            y = y[0::2, 0::2] if stride == 2 else y
        else:
            assert stride == 1
            y = (xb.reshape(-1, Cin) @ W.reshape(Cout, Cin).T).reshape(xb.shape[0], xb.shape[1], Cout)
        y = y.reshape(Ho * Wo, Cout).T
        if bias:
            y = y + dev(bias, np.float32, Cout)[:, None]
        if addbc:
            y = y + dev(addbc + 4 * b * Cout, np.float32, Cout)[:, None]
        if resid:
            y = y + dev(resid + 4 * b * Cout * Ho * Wo, np.float32, Cout * Ho * Wo).reshape(Cout, Ho * Wo)
        o[b] = y


def k_gn_f32(L):
    A = L["args"]
    x, gamma, beta, y, Cc, HW, eps, act = u64(A[0]), u64(A[1]), u64(A[2]), u64(A[3]), i32(A[4]), i32(A[5]), f32(A[6]), i32(A[7])
    B = int(L["g"].split(",")[1])
    v = dev(x, np.float32, B * Cc * HW).reshape(B, 32, Cc // 32, HW).astype(np.float64)
    mean, var = v.mean(axis=(2, 3), keepdims=True), v.var(axis=(2, 3), keepdims=True)
    r = ((v - mean) / np.sqrt(var + eps)).reshape(B, Cc, HW).astype(np.float32) * dev(gamma, np.float32, Cc)[:, None] + dev(beta, np.float32, Cc)[:, None]
    dev(y, np.float32, B * Cc * HW)[:] = (silu(r) if act else r).astype(np.float32).ravel()


def k_linear_f32(L):
    A = L["args"]
    x, w, bias, out, I, O, silu_in = u64(A[0]), u64(A[1]), u64(A[2]), u64(A[3]), i32(A[4]), i32(A[5]), i32(A[6])
    B = int(L["g"].split(",")[1])
    v = dev(x, np.float32, B * I).reshape(B, I)
    v = silu(v) if silu_in else v
    dev(out, np.float32, B * O)[:] = (v @ dev(w, np.float32, O * I).reshape(O, I).T + dev(bias, np.float32, O)).astype(np.float32).ravel()


def k_timestep_f32(L):
    A = L["args"]
    t, out, dim = u64(A[0]), u64(A[1]), i32(A[2])
    B = int(L["g"].split(",")[0])
    half = dim // 2
    ang = dev(t, np.float32, B).astype(np.float64)[:, None] * np.exp(-np.log(10000.0) * np.arange(half) / half)[None]
    dev(out, np.float32, B * dim)[:] = np.concatenate([np.cos(ang), np.sin(ang)], 1).astype(np.float32).ravel()


def k_attn_f32(L):
    A = L["args"]
    q, k, v, out, Cc, T = u64(A[0]), u64(A[1]), u64(A[2]), u64(A[3]), i32(A[4]), i32(A[5])
    B = int(L["g"].split(",")[2])
    rd = lambda p: dev(p, np.float32, B * Cc * T).reshape(B, Cc // 8, 8, T)
    sc = np.einsum("bhdt,bhds->bhts", rd(q), rd(k)) * 0.35355339059327373
    p = np.exp(sc - sc.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    dev(out, np.float32, B * Cc * T)[:] = np.einsum("bhts,bhds->bhdt", p, rd(v)).astype(np.float32).ravel()


def k_put_channels_f32(L):
    A = L["args"]
    src, dst, Cc, Ctot, c_off, HW, total = u64(A[0]), u64(A[1]), i32(A[2]), i32(A[3]), i32(A[4]), i32(A[5]), u64(A[6])
    B = total // (Cc * HW)
    d_ = np.lib.stride_tricks.as_strided(dev(dst + 4 * c_off * HW, np.float32, (B - 1) * Ctot * HW + Cc * HW), shape=(B, Cc * HW),
                                         strides=(4 * Ctot * HW, 4))
    d_[:] = dev(src, np.float32, total).reshape(B, Cc * HW)


KERNELS = {"conv_f32_kernel": k_conv_f32, "gn_f32_kernel": k_gn_f32, "linear_f32_kernel": k_linear_f32, "timestep_f32_kernel": k_timestep_f32,
           "attn_f32_kernel": k_attn_f32, "put_channels_f32_kernel": k_put_channels_f32, "attention_kernel": k_attention, "pointwise_f32_kernel": k_pointwise_f32, "gn_apply_kernel": k_gn_apply, "softmax_rows_kernel": k_softmax_rows,
           "temb_mlp_kernel": k_temb_mlp, "conv_igemm": k_igemm, "splitk_reduce_kernel": k_splitk_reduce, "conv_in_kernel": k_conv_in,
           "gn_stats_kernel": k_gn_stats, "gn_small_kernel": k_gn_small, "gn_finalize2_kernel": k_gn_finalize2, "conv_t32": k_conv_t32, "conv_s": k_conv_s,
           "iadb_step_kernel": k_iadb_step, "ddim_step_kernel": k_ddim_step, "head_conv": k_head_conv}

# what is run: (in, out channels, resolution, block_out_channels / attention levels, batch, mode)
CASES = {
    "c2": (3, 6, 64, drive.RES64, 2, "forward"),           # cat_res64, UNet 3 -> 6
    "c2loop": (3, 6, 64, drive.RES64, 2, "iadb"),          # ... two steps of the in-engine IADB loop with snapshots
    "c3loop": (3, 3, 64, drive.RES64, 1, "ddim"),          # church_res64: two steps of the in-engine DDIM loop
    "c4": (3, 6, 128, drive.RES128, 1, "forward"),         # celeba_res128
    "c5": (4, 8, 64, drive.RES64, 1, "forward"),           # latent UNet 4 -> 8
    "cond": (6, 3, 64, drive.RES64, 1, "cond"),            # super-resolution sampler: x (3) + conditioning (3) -> 3, two steps
                                                           # (the reference runs it at 128 px; the 128-px layout itself is case c4)
    "vae16": (4, 3, 16, None, 1, "vae"),                   # AutoencoderKL decoder, full layout, 16x16 latent -> 128 px
    # layouts the reference ships beyond the benchmark configurations (tests/test_gpu_layouts.py) and first-level widths 64 / 256
    "lat256": (4, 8, 32, ((128, 256, 256), 2, 0), 5, "forward"),       # latent celeba_res256: 64-token attention, ragged batch
    "w64": (3, 6, 64, ((64, 128, 128), 2, 0), 3, "forward"),           # groups of two channels; 64 + 128 = 192-channel concats
    "w256": (3, 6, 32, ((256, 256, 512), 2, 0), 1, "forward"),
    "w64x4": (3, 6, 64, ((64, 64, 128, 256), 3, 0), 2, "forward"),
    "deep32": (3, 6, 32, ((128, 128, 256, 256, 512), 3, 1), 1, "forward"),   # five levels 32 .. 2 px, 4x4 attention: every kernel family of c2 at 1/4 of its size
    "lat256f32": (4, 8, 32, ((128, 256, 256), 2, 0), 1, "forward"),    # the fp32-compute mode at a size the instruction-level simulator can afford      # tests/test_gpu_first_level_widths.py's second layout: 8x8 attention
    "bottom1x1": (3, 6, 32, drive.RES64, 3, "forward"),    # the res64 layout on 32x32 inputs: last level 1x1, deferred split-K in
                                                           # front of a conv_s upsampler (round-3 advisor finding), ragged batch 3
    "c2bf16": (3, 6, 64, drive.RES64, 1, "forward"),       # bf16 storage / MFMA inputs (case name ends in bf16)
    "c2f32": (3, 6, 64, drive.RES64, 2, "forward"),        # the fp32-compute verification mode (case name ends in f32)
}
# kernel-coverage cases of tests/gfx950sim/suite.py (variants no benchmark configuration launches): simulator only, not replayed by
# tests/test_launch_trace.py
SIM_CASES = {
    "big128": (3, 6, 64, ((128, 128), 9, 9), 12, "forward"),           # with BNDM_NO_FUSED: 49152 x 128 outputs = 192 tiles of 256 x 128 (conv_igemm's 3-stage tile)
    "big128bf16": (3, 6, 64, ((128, 128), 9, 9), 12, "forward"),
    "w64bf16": (3, 6, 64, ((64, 128, 128), 2, 0), 1, "forward"),       # bf16 instances of the generic kernels (gn_apply, attention, igemm head)
    "lat256bf16": (4, 8, 32, ((128, 256, 256), 2, 0), 1, "forward"),   # bf16 conv_t32<TH=8> incl. the 32-channel head
    "vae16bf16": (4, 3, 16, None, 1, "vae"),                           # bf16 softmax_rows (256-token attention of the VAE mid block)
    "loop16f32": (4, 8, 16, ((128, 128), 9, 9), 1, "iadb"),                # the in-engine loop in fp32 mode (fill_f32_kernel)
}
T_IN, DA, DG = [1.0, 0.5], [-0.5, -0.5], [-0.3, -0.2]
DDIM = [990.0, 0.9, 0.43588989, 0.92, 0.39191836, 980.0, 0.92, 0.39191836, 0.94, 0.34117444]


def main():
    libpath, outdir, case = os.path.abspath(sys.argv[1]), sys.argv[2], sys.argv[3]
    cin, cout, res, layout, B, mode = {**CASES, **SIM_CASES}[case]
    B = int(os.environ.get("EXEC_BATCH", B))                # (experiments: another batch for the same case)
    MB = int(os.environ.get("EXEC_MAX_BATCH", B))            # handle sized for a larger batch than the call's (tile choices follow it)
    global BF16
    BF16 = case.endswith("bf16")
    dtype = drive.F32 if case.endswith("f32") else (drive.BF16 if BF16 else drive.F16)
    _lib.LIB_PATH = libpath
    lib = _lib.load()
    d = drive.Dev()
    rs = np.random.RandomState(21)
    h = C.c_void_p()
    if mode == "vae":
        cfg = _lib.VaeConfig()
        cfg.latent_channels, cfg.out_channels, cfg.latent_resolution, cfg.num_levels = cin, cout, res, 4
        for i, v in enumerate((128, 256, 512, 512)):
            cfg.block_out_channels[i] = v
        cfg.layers_per_block, cfg.dtype, cfg.max_batch = 2, dtype, MB
        _lib.check(lib.bndm_vae_decoder_create(C.byref(h), C.byref(cfg)), "vae create")
    else:
        cfg = drive.unet_cfg(cin, cout, res, *layout, dtype, MB)
        _lib.check(lib.bndm_unet_create(C.byref(h), C.byref(cfg)), "create")
    name, numel = C.create_string_buffer(200), C.c_int64()
    given = np.load(sys.argv[4])                           # state dict written by the test (the oracle's initialisation)
    for i in range(lib.bndm_unet_num_params(h)):
        _lib.check(lib.bndm_unet_param_info(h, i, name, 200, C.byref(numel)), "param_info")
        w = np.ascontiguousarray(given[name.value.decode()], np.float32).ravel()
        assert w.size == numel.value, name.value
        _lib.check(lib.bndm_unet_load_param(h, name.value, w.ctypes.data_as(C.c_void_p), numel.value), "load_param")
    lanes = int(os.environ.get("EXEC_LANES", "1"))           # candidate (tools/experiments/lanes.patch): chains of launches
    if lanes > 1:
        lib.bndm_unet_set_lanes.restype, lib.bndm_unet_set_lanes.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_int]
        _lib.check(lib.bndm_unet_set_lanes(h, lanes, int(os.environ.get("EXEC_LANE_FLAGS", "0"))), "set_lanes")
    _lib.check(lib.bndm_unet_finalize(h), "finalize")
    cx = 3 if mode == "cond" else cin                      # channels of the sampler state
    x = d.alloc(B * cx * res * res * 4)
    xin = rs.standard_normal((B, cx, res, res)).astype(np.float32)
    dev(x.value, np.float32, xin.size)[:] = xin.ravel()
    np.save(os.path.join(outdir, f"exec_{case}_x.npy"), xin)
    sim = None
    if os.environ.get("EXEC_SIM") in ("1", "diff"):
        # tests/gfx950sim: the library's REAL machine code runs, launch by launch, on the instruction-level simulator
        # (the numpy contract models below are not involved)
        from tests.gfx950sim.runtime import Simulator
        sim = Simulator(libpath, verbose=os.environ.get("EXEC_SIM_VERBOSE") == "1").install()
        if os.environ["EXEC_SIM"] == "diff":
            # launch by launch: the simulated kernel's stores against the numpy contract model of the same launch
            def reference(name, grid, block, lds, raw):
                KERNELS[H.short_name(name)](dict(sym=name, name=H.short_name(name), g=",".join(map(str, grid)), b=",".join(map(str, block)),
                                                 lds=str(lds), args=list(raw)))
            sim.reference = reference
    drive.mark("run")
    if mode == "forward":
        t, o = d.alloc(B * 4), d.alloc(B * cout * res * res * 4)
        tin = np.linspace(0.15, 0.85, B).astype(np.float32)
        dev(t.value, np.float32, B)[:] = tin
        np.save(os.path.join(outdir, f"exec_{case}_t.npy"), tin)
        _lib.check(lib.bndm_unet_forward(h, x, t, o, B, None), "forward")
        result = lambda: dev(o.value, np.float32, B * cout * res * res).reshape(B, cout, res, res).copy()
    elif mode == "vae":
        o = d.alloc(B * cout * (8 * res) ** 2 * 4)
        _lib.check(lib.bndm_vae_decode(h, x, o, B, None), "vae_decode")
        result = lambda: dev(o.value, np.float32, B * cout * (8 * res) ** 2).reshape(B, cout, 8 * res, 8 * res).copy()
    elif mode in ("iadb", "cond"):
        extra = None
        if mode == "cond":
            extra = d.alloc(B * 3 * res * res * 4)
            ein = rs.standard_normal((B, 3, res, res)).astype(np.float32)
            dev(extra.value, np.float32, ein.size)[:] = ein.ravel()
            np.save(os.path.join(outdir, f"exec_{case}_extra.npy"), ein)
        snaps = d.alloc(2 * B * cx * res * res * 4)
        _lib.check(lib.bndm_unet_sample_iadb(h, x, extra, B, cx, 2, drive.farr(T_IN), drive.farr(DA), drive.farr(DG),
                                             (C.c_uint8 * 2)(1, 1), snaps, None), "sample_iadb")
        result = lambda: dev(snaps.value, np.float32, 2 * B * cx * res * res).reshape(2, B, cx, res, res).copy()
    else:
        _lib.check(lib.bndm_unet_sample_ddim(h, x, B, 2, drive.farr(DDIM), 1.0, None), "sample_ddim")
        result = lambda: dev(x.value, np.float32, B * cx * res * res).reshape(B, cx, res, res).copy()
    d.flush()
    if sim is not None:
        sim.uninstall()
        out = result()
        np.save(os.path.join(outdir, f"exec_{case}_out.npy"), out)
        fam = {}
        for name, grid, ninst, dt, nhz in sim.log:
            f = fam.setdefault(H.short_name(name), [0, 0, 0.0])
            f[0] += 1
            f[1] += ninst
            f[2] += dt
        print(f"OK simulated {len(sim.log)} launches; output rms {float(np.sqrt((out ** 2).mean())):.4f}; hazards {len(sim.hazards)}")
        for k, (n_, ni, dt) in sorted(fam.items()):
            print(f"   {k:24s} {n_:4d} launches {ni:11d} wave-instructions {dt:8.1f} s")
        for hz in sim.hazards[:20]:
            print("HAZARD", hz)
        if sim.stats:                                          # GFX950SIM_STATS=1 (single process): executed instructions by mnemonic
            json.dump([dict(kernel=n_, grid=g_, insts=st_, bytes=by_) for n_, g_, st_, by_ in sim.stats],
                      open(os.path.join(outdir, f"sim_stats_{case}.json"), "w"))
        return
    dev(x.value, np.float32, xin.size)[:] = xin.ravel()    # (nothing ran: the state is still x0; written again for clarity)
    lines = open(os.environ["HIPMOCK_TRACE"]).read().splitlines()
    n = 0
    for ln in dict(H.stages(lines))["run"]:
        if ln.startswith("launch "):
            L = H.parse_launch(ln)
            KERNELS[L["name"]](L)
            n += 1
        elif ln.startswith("memcpy_async") and "src=0x" in ln:             # device -> device (snapshots): at its place in the order
            m = re.search(r"dst=(0x[0-9a-f]+) src=(0x[0-9a-f]+) n=(\d+)", ln)
            dst, src, nb = int(m.group(1), 16), int(m.group(2), 16), int(m.group(3))
            dev(dst, np.uint8, nb)[:] = dev(src, np.uint8, nb)
        else:
            # host -> device copies (schedule tables) were carried out when they were recorded; allocations, attributes, events
            assert ln.startswith(("==", "event_record", "funcattr", "memcpy", "malloc", "free", "hostmalloc", "stream_", "device_sync")), \
                f"a call the replay does not model: {ln[:80]}"
    out = result()
    np.save(os.path.join(outdir, f"exec_{case}_out.npy"), out)
    kinds = set()
    kern, lab, fl = C.create_string_buffer(128), C.create_string_buffer(256), C.c_double()
    for i in range(lib.bndm_unet_num_ops(h)):
        _lib.check(lib.bndm_unet_op_info(h, i, kern, 128, lab, 256, C.byref(fl)), "op_info")
        kinds.add(kern.value.decode())
    print(f"OK replayed {n} launches; output rms {float(np.sqrt((out ** 2).mean())):.4f}; kernels {sorted(kinds)}")


if __name__ == "__main__":
    main()
