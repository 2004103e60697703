"""TEST INFRASTRUCTURE: what the engine asks conv_s (the <= 8x8 levels, csrc/unet_tail.hip) to compute, checked on the CPU
against the layers' definition.  Companion of check_conv_t32.py; same method, under the recording HIP runtime:

  * sources, residual and additive rows of every conv_s launch of one forward (res64, 3 -> 6, B = 1) are filled with random data;
  * MODEL: the launch evaluated from its kernel arguments and uploaded tables alone -- the per-wave step lists (`desc`: table
    slot = tap shift, 32-channel sub-chunk, round boundaries, no-work entries), the round table (source tensor, channel window,
    same-resolution / nearest-2x / stride-2 phase row maps), the weight stream in consumption order ([n-tile][wave][entry]
    [ks * NB + nb][lane][8]), bias + time-embedding rows, residual, the q|k|v + softmax epilogue, and the GroupNorm(+SiLU)
    copies requested for the consumers (group size, gamma / beta);
  * REFERENCE: the layer from the state dict's ORIGINAL tensors by the module's definition (ResnetBlock2D convs and shortcuts,
    Downsample2D stride 2 / pad 1, Upsample2D nearest 2x + conv, Attention to_q / to_k / to_v / softmax / to_out), fp32; for
    the normalised copies: GroupNorm with the consumer's parameters, which are FOUND by value in the state dict (the launch only
    carries pointers) and must be a norm layer whose group size matches the request.

Says that the step lists, round tables, weight stream, row maps and consumer requests are right; says nothing about device code.
"""
import ctypes as C
import os
import re
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from bndm_amd import _lib  # noqa: E402
from tests.hipmock import drive, harness as H  # noqa: E402
from tests.hipmock.check_conv_t32 import conv3x3, dev, silu, up2  # noqa: E402


class Round(C.Structure):
    _fields_ = [("src", C.c_uint64), ("row_bytes", C.c_int), ("cbyte", C.c_int), ("mode", C.c_int), ("phase", C.c_int),
                ("nsub", C.c_int), ("pad", C.c_int)]


class Norm(C.Structure):
    _fields_ = [("out", C.c_uint64), ("gamma", C.c_uint64), ("beta", C.c_uint64), ("gs", C.c_int), ("silu", C.c_int)]


class TailArgs(C.Structure):                       # csrc/unet_kernels.hpp: struct TailArgs
    _fields_ = [("wgt", C.c_uint64), ("desc", C.c_uint64), ("rounds", C.c_uint64), ("r0", Round), ("r1", Round),
                ("nrounds", C.c_int), ("maxsteps", C.c_int), ("nuse", C.c_int * 8), ("tile_bytes", C.c_longlong),
                ("wave_bytes", C.c_int), ("B", C.c_int), ("hwlog", C.c_int), ("wlog", C.c_int), ("Cout", C.c_int),
                ("ntn", C.c_int), ("bias", C.c_uint64), ("temb", C.c_uint64), ("temb_bstride", C.c_int),
                ("temb_off", C.c_int), ("resid", C.c_uint64), ("raw_out", C.c_uint64), ("nreq", C.c_int), ("req", Norm * 3),
                ("eps", C.c_float), ("epi", C.c_int), ("attn_out", C.c_uint64), ("dbg", C.c_uint64)]


assert C.sizeof(TailArgs) == 328 and TailArgs.req.offset == 208 and TailArgs.eps.offset == 304


def rel(a, b):
    return float(np.linalg.norm((a - b).astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))


def gn_rows(v, gs, HW, gamma, beta, eps, act):
    """v [M, C] fp32; groups of gs channels x the HW rows of a sample"""
    M, Cc = v.shape
    g = v.reshape(M // HW, HW, Cc // gs, gs).astype(np.float64)
    mean = g.mean(axis=(1, 3), keepdims=True)
    var = ((g - mean) ** 2).mean(axis=(1, 3), keepdims=True)
    y = ((g - mean) / np.sqrt(var + eps)).reshape(M, Cc).astype(np.float32) * gamma + beta
    return silu(y) if act else y


def main():
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
    lib = _lib.load()
    d = drive.Dev()
    rs = np.random.RandomState(5)
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
    norms = {k: v for k, v in sd.items() if k.endswith(".weight") and "norm" in k.split(".")[-2]}
    drive.mark("forward")
    drive.forward(lib, d, h, 1, 3, 6, 64)
    d.flush()
    lines = open(os.environ["HIPMOCK_TRACE"]).read().splitlines()
    launches = [H.parse_launch(x) for x in dict(H.stages(lines))["forward"] if x.startswith("launch ")]
    tl = [x for x in launches if x["name"] == "conv_s"]
    kern, lab, fl = C.create_string_buffer(128), C.create_string_buffer(256), C.c_double()
    labels = []
    for i in range(lib.bndm_unet_num_ops(h)):
        _lib.check(lib.bndm_unet_op_info(h, i, kern, 128, lab, 256, C.byref(fl)), "op_info")
        if kern.value.startswith(b"conv_s"):
            labels.append(" ".join(lab.value.decode().split()))
    assert len(labels) == len(tl), (len(labels), len(tl))

    worst = 0.0
    for label, ln in zip(labels, tl):
        a = TailArgs.from_buffer_copy(ln["args"][0])
        TM, NB, D = map(int, re.search(r"conv_sID[^_]*_?Li(\d+)ELi(\d+)ELi(\d+)E", ln["sym"]).groups())
        mod = label.split()[1]
        HW, Wd = 1 << a.hwlog, 1 << a.wlog
        Hd = HW // Wd
        M = a.B * HW
        TN, NL = NB * 32, 2 * NB
        attn = a.epi == 1
        ncol = a.ntn * TN                                                      # accumulator columns over all n-tiles
        # ---- rounds, and the tensors they read -----------------------------------------------------------------
        rt = [Round.from_buffer_copy(bytes(dev(a.rounds + 32 * r, np.uint8, 32))) for r in range(a.nrounds)]
        assert bytes(rt[0]) == bytes(a.r0) and (a.nrounds < 2 or bytes(rt[1]) == bytes(a.r1)), label
        srcs = {}
        for r in rt:
            if r.src not in srcs:
                Cs = r.row_bytes // 2
                rows_src = M if r.mode == 0 else (M >> 2 if r.mode == 1 else M << 2)
                buf = dev(r.src, np.float16, rows_src * Cs)
                buf[:] = rs.standard_normal(buf.size).astype(np.float16)
                srcs[r.src] = buf.astype(np.float32).reshape(rows_src, Cs)

        def rows_of(r):
            """[M, C_src] activation rows as the round's row map presents them to the output grid"""
            x = srcs[r.src]
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
            ys, xs_ = slice(max(0, -dy), Hd - max(0, dy)), slice(max(0, -dx), Wd - max(0, dx))
            yd, xd = slice(max(0, dy), Hd - max(0, -dy)), slice(max(0, dx), Wd - max(0, -dx))
            out[:, ys, xs_] = g[:, yd, xd]
            return out.reshape(M, -1)

        # ---- MODEL: walk every wave's step list exactly as the kernel does --------------------------------------
        desc = dev(a.desc, np.uint32, 8 * a.maxsteps).reshape(8, a.maxsteps)
        acc = np.zeros((M, ncol), np.float32)
        Wdense = {}                                       # (round, table slot) -> [ncol][256] weights of that step set
        taps_of_round = {}
        for w in range(8):
            rnd = 0
            for s in range(a.maxsteps):
                e = int(desc[w, s])
                if not (e & 0x2000):
                    ts, j = e & 15, (e >> 4) & 7
                    assert ts < 9 and ((e >> 7) & 1) == (rnd & 1) and j < rt[rnd].nsub and s < a.nuse[w], (label, w, s, hex(e))
                    taps_of_round.setdefault(rnd, set()).add(ts)
                    Wt = Wdense.setdefault((rnd, ts), np.zeros((ncol, 256), np.float32))
                    for nt in range(a.ntn):
                        frag = dev(a.wgt + nt * a.tile_bytes + w * a.wave_bytes + s * NL * 1024, np.float16, NL * 512)
                        frag = frag.astype(np.float32).reshape(2, NB, 64, 8)                 # [ks][nb][lane][8]
                        for ks in range(2):
                            for nb in range(NB):
                                for half in range(2):                                        # lane >> 5
                                    c = 32 * j + 16 * ks + 8 * half
                                    Wt[nt * TN + nb * 32:nt * TN + nb * 32 + 32, c:c + 8] += frag[ks, nb, 32 * half:32 * half + 32]
                if e & 0x100:
                    rnd += 1
            assert rnd == a.nrounds - 1, (label, w, rnd)
        for (rnd, ts), Wt in Wdense.items():
            r = rt[rnd]
            A = rows_of(r)[:, r.cbyte // 2:r.cbyte // 2 + 32 * r.nsub]
            acc += shifted(A, ts) @ Wt[:, :32 * r.nsub].T
        col = np.arange(ncol)
        nt_, c_ = col // TN, col % TN
        cb = np.where(attn, (c_ >> 5) * a.Cout + nt_ * 32 + (c_ & 31), col) if attn else col      # channel of accumulator column
        add = np.zeros(ncol, np.float32)
        nchan = 3 * a.Cout if attn else a.Cout
        if a.bias:
            add += dev(a.bias, np.float32, nchan)[cb]
        if a.temb:
            trow = dev(a.temb + 4 * a.temb_off, np.float32, nchan)
            trow[:] = rs.standard_normal(nchan).astype(np.float32)
            add += trow[cb]
        v = acc + add
        resid = None
        if a.resid:
            rb = dev(a.resid, np.float16, M * a.Cout)
            rb[:] = rs.standard_normal(rb.size).astype(np.float16)
            resid = rb.astype(np.float32).reshape(M, a.Cout)
            v = v + resid
        if attn:
            qkv = v.reshape(M, a.ntn, 3, 4, 8)                                              # [row][n-tile][q|k|v][head][8]
            q, k, vv = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
            qs = q.reshape(a.B, HW, a.ntn * 4, 8) * 0.35355339059327373
            sc = np.einsum("bthe,bshe->bhts", qs, k.reshape(a.B, HW, a.ntn * 4, 8))
            p = np.exp(sc - sc.max(-1, keepdims=True))
            p /= p.sum(-1, keepdims=True)
            model = np.einsum("bhts,bshe->bthe", p, vv.reshape(a.B, HW, a.ntn * 4, 8)).reshape(M, a.Cout)
        else:
            model = v[:, :a.Cout]

        # ---- REFERENCE -------------------------------------------------------------------------------------------
        order = []                                         # distinct sources in round order, split into 3x3-like and 1x1 sets
        for i, r in enumerate(rt):
            one = taps_of_round.get(i, {4}) == {4} and r.mode == 0
            if not order or order[-1][0] != r.src:
                order.append([r.src, one, r.mode])
        def P(k):
            return sd[k]
        def grid(x, C_, hh, ww):
            return x.reshape(hh, ww, C_)
        if mod.endswith("qkv+attn"):
            blk = mod[:-len(".qkv+attn")]
            x = srcs[order[0][0]]
            Cc = a.Cout
            qr = x @ P(blk + ".to_q.weight").reshape(Cc, Cc).T + P(blk + ".to_q.bias")
            kr = x @ P(blk + ".to_k.weight").reshape(Cc, Cc).T + P(blk + ".to_k.bias")
            vr = x @ P(blk + ".to_v.weight").reshape(Cc, Cc).T + P(blk + ".to_v.bias")
            nh = Cc // 8
            sc = np.einsum("the,she->hts", qr.reshape(HW, nh, 8), kr.reshape(HW, nh, 8)) / np.sqrt(8.0)
            p = np.exp(sc - sc.max(-1, keepdims=True))
            p /= p.sum(-1, keepdims=True)
            ref = np.einsum("hts,she->the", p, vr.reshape(HW, nh, 8)).reshape(M, Cc)
        elif mod.endswith("to_out"):
            blk = mod[:-len(".to_out")]
            ref = srcs[order[0][0]] @ P(blk + ".to_out.0.weight").reshape(a.Cout, -1).T + P(blk + ".to_out.0.bias") + resid
        else:
            three = [o for o in order if not o[1]]
            ones = [o for o in order if o[1]]
            if mod.endswith("downsamplers.0.conv"):
                x = srcs[three[0][0]].reshape(2 * Hd, 2 * Wd, -1)
                wgt = P(mod + ".weight").reshape(a.Cout, -1, 3, 3)
                full = conv3x3(x, wgt)                      # stride 1, pad 1 ...
                ref = full[0::2, 0::2].reshape(M, -1) + P(mod + ".bias")      # ... sampled at the stride-2 positions
            else:
                parts = []
                for src, _, mode in three:
                    x = srcs[src]
                    parts.append(up2(x.reshape(Hd // 2, Wd // 2, -1)) if mode == 1 else x.reshape(Hd, Wd, -1))
                cat = np.concatenate(parts, -1)
                if mod.endswith("upsamplers.0.conv"):
                    ref = conv3x3(cat, P(mod + ".weight").reshape(a.Cout, -1, 3, 3)).reshape(M, -1) + P(mod + ".bias")
                elif mod.endswith(".conv1"):
                    ref = conv3x3(cat, P(mod + ".weight").reshape(a.Cout, -1, 3, 3)).reshape(M, -1) + add[:a.Cout]
                    if a.bias:                               # (a separate bias would have to be the module's)
                        assert np.array_equal(dev(a.bias, np.float32, a.Cout), P(mod + ".bias")), label
                else:
                    blk = re.sub(r"\.conv2(\+sc)?$", "", mod)
                    ref = conv3x3(cat, P(blk + ".conv2.weight").reshape(a.Cout, -1, 3, 3)).reshape(M, -1) + P(blk + ".conv2.bias")
                    if mod.endswith("+sc"):
                        xin = np.concatenate([srcs[o[0]] for o in ones], -1)
                        ref = ref + xin @ P(blk + ".conv_shortcut.weight").reshape(a.Cout, -1).T + P(blk + ".conv_shortcut.bias")
                    else:
                        assert not ones and resid is not None, label
                        ref = ref + resid
        err = rel(model, ref)
        # ---- consumers: GroupNorm(+SiLU) copies requested from this launch ---------------------------------------
        cons = []
        for q in range(a.nreq):
            rq = a.req[q]
            gam, bet = dev(rq.gamma, np.float32, a.Cout), dev(rq.beta, np.float32, a.Cout)
            found = None
            for kname, arr in norms.items():
                for off in range(0, arr.size - a.Cout + 1, 32):
                    if np.array_equal(arr[off:off + a.Cout], gam):
                        found = (kname, off, arr.size)
            assert found, f"{label}: request {q}: gamma is no slice of a norm layer's weight"
            kname, off, total = found
            assert np.array_equal(sd[kname[:-len('weight')] + "bias"][off:off + a.Cout], bet), (label, kname)
            assert rq.gs == total // 32, f"{label}: {kname} has groups of {total // 32} channels, request says {rq.gs}"
            assert off % rq.gs == 0 and a.Cout % rq.gs == 0, (label, kname, off)          # whole groups inside this tensor
            copy_model = gn_rows(v[:, :a.Cout], rq.gs, HW, gam, bet, a.eps, rq.silu)
            copy_ref = gn_rows(ref, rq.gs, HW, sd[kname][off:off + a.Cout], sd[kname[:-len('weight')] + "bias"][off:off + a.Cout], 1e-5, "group_norm" not in kname)
            err = max(err, rel(copy_model, copy_ref))
            cons.append(f"{kname[:-7]}[{off}:{off + a.Cout}]" + ("" if rq.silu else " (no SiLU)"))
        worst = max(worst, err)
        print(f"{err:.2e}  {label}  -> {', '.join(cons) if cons else 'raw only'}")
        assert err <= 5e-3, f"{label}: rel-L2 {err:.3e}"
    print(f"OK {len(tl)} conv_s launches, worst rel-L2 {worst:.2e}")


if __name__ == "__main__":
    main()
