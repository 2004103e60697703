"""TEST INFRASTRUCTURE: drives a libbndm_hip.so through its C ABI (ctypes + numpy only, no torch) under the recording HIP
stand-in of tests/hipmock/hipmock.cpp, so that the library's HOST side -- launch lists, grids, kernel arguments, buffer
layout, table uploads -- leaves a trace that can be compared between two builds or against a committed digest.

    LD_LIBRARY_PATH=<dir with the stand-in's libamdhip64.so.7> HIPMOCK_TRACE=trace.txt HIPMOCK_KERNARGS=ka.txt \
        python tests/hipmock/drive.py <lib.so> <scenario> [lanes [flags]]

Scenarios mirror BASELINE.json's configurations at the sizes bench.py and tests/test_gpu_benched.py run them.
A line "== <mark>" is written into the trace between the stages of a scenario."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bndm_amd import _lib  # noqa: E402

F16, BF16, F32 = 0, 1, 2


def mark(text):
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipmock_mark(text.encode())


class Dev:
    """device memory of the stand-in (hipMalloc through the runtime the library itself is linked to)"""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so.7")
        assert hasattr(self.hip, "hipmock_flush"), "the real HIP runtime is loaded: run under tests/hipmock's stand-in"

    def alloc(self, nbytes):
        p = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
        return p

    def flush(self):
        self.hip.hipmock_flush()


def unet_cfg(cin, cout, res, boc, down_attn, up_attn, dtype, max_batch):
    cfg = _lib.UNetConfig()
    cfg.in_channels, cfg.out_channels, cfg.resolution, cfg.num_levels = cin, cout, res, len(boc)
    for i, v in enumerate(boc):
        cfg.block_out_channels[i] = v
        cfg.down_attn[i] = int(i == down_attn)
        cfg.up_attn[i] = int(i == up_attn)
    cfg.layers_per_block, cfg.dtype, cfg.max_batch = 2, dtype, max_batch
    return cfg


def load_params(lib, h, seed):
    rs = np.random.RandomState(seed)
    name = C.create_string_buffer(200)
    numel = C.c_int64()
    for i in range(lib.bndm_unet_num_params(h)):
        _lib.check(lib.bndm_unet_param_info(h, i, name, 200, C.byref(numel)), "param_info")
        w = (rs.standard_normal(numel.value) * 0.05).astype(np.float32)
        if name.value.decode().split(".")[-2].startswith(("norm", "group_norm", "conv_norm")) and name.value.endswith(b"weight"):
            w += 1.0
        _lib.check(lib.bndm_unet_load_param(h, name.value, w.ctypes.data_as(C.c_void_p), numel.value), "load_param")


def make_unet(lib, cfg, lanes=1, flags=0):
    h = C.c_void_p()
    _lib.check(lib.bndm_unet_create(C.byref(h), C.byref(cfg)), "create")
    load_params(lib, h, 0)
    if lanes > 1:
        _lib.check(lib.bndm_unet_set_lanes(h, lanes, flags), "set_lanes")
    mark("finalize")
    _lib.check(lib.bndm_unet_finalize(h), "finalize")
    return h


def farr(v):
    return (C.c_float * len(v))(*v)


RES64 = ((128, 128, 256, 256, 512, 512), 4, 1)
RES128 = ((128, 128, 128, 256, 256, 512, 512), 5, 1)


def iadb(lib, dev, h, B, C_, cin, res, steps, snap=None, cond=False):
    x = dev.alloc(B * C_ * res * res * 4)
    extra = dev.alloc(B * (cin - C_) * res * res * 4) if cond else None
    t = farr([(steps - s) / steps for s in range(steps)])
    da = farr([-1.0 / steps] * steps)
    dg = farr([-0.5 / steps] * steps)
    mask = snaps = None
    if snap:
        mask = (C.c_uint8 * steps)(*snap)
        snaps = dev.alloc(sum(snap) * B * C_ * res * res * 4)
    _lib.check(lib.bndm_unet_sample_iadb(h, x, extra, B, C_, steps, t, da, dg, mask, snaps, None), "sample_iadb")


def forward(lib, dev, h, B, cin, cout, res):
    x = dev.alloc(B * cin * res * res * 4)
    t = dev.alloc(B * 4)
    o = dev.alloc(B * cout * res * res * 4)
    _lib.check(lib.bndm_unet_forward(h, x, t, o, B, None), "forward")


def scenario(lib, dev, name, lanes, flags):
    if name == "c2":                                       # cat_res64 IADB, B = 64, UNet 3 -> 6
        h = make_unet(lib, unet_cfg(3, 6, 64, *RES64, F16, 64), lanes, flags)
        mark("forward B=64")
        forward(lib, dev, h, 64, 3, 6, 64)
        mark("sample_iadb B=64 steps=3 snapshots at 1,2")
        iadb(lib, dev, h, 64, 3, 3, 64, 3, snap=[0, 1, 1])
        mark("forward B=2 on the B=64 handle")
        forward(lib, dev, h, 2, 3, 6, 64)
        mark("sample_iadb B=6 steps=2")
        iadb(lib, dev, h, 6, 3, 3, 64, 2)
    elif name == "b500":                                   # the reference's shipped batch at 64 px (scripts/sampling/cat_res64_test.sh:5)
        h = make_unet(lib, unet_cfg(3, 6, 64, *RES64, F16, 500), lanes, flags)
        mark("sample_iadb B=500 steps=1")
        iadb(lib, dev, h, 500, 3, 3, 64, 1)
    elif name == "b200":                                   # ... and at 128 px (cat_res128_test.sh:4)
        h = make_unet(lib, unet_cfg(3, 6, 128, *RES128, F16, 200), lanes, flags)
        mark("sample_iadb B=200 steps=1")
        iadb(lib, dev, h, 200, 3, 3, 128, 1)
    elif name == "c2bf16":
        h = make_unet(lib, unet_cfg(3, 6, 64, *RES64, BF16, 64), lanes, flags)
        mark("forward B=64")
        forward(lib, dev, h, 64, 3, 6, 64)
    elif name == "c3":                                     # church_res64 DDIM, B = 64, UNet 3 -> 3
        h = make_unet(lib, unet_cfg(3, 3, 64, *RES64, F16, 64), lanes, flags)
        x = dev.alloc(64 * 3 * 64 * 64 * 4)
        coef = farr([990.0, 0.9, 0.43, 0.92, 0.39, 980.0, 0.92, 0.39, 0.94, 0.34])
        mark("sample_ddim B=64 steps=2")
        _lib.check(lib.bndm_unet_sample_ddim(h, x, 64, 2, coef, 1.0, None), "sample_ddim")
    elif name == "c4":                                     # celeba_res128 IADB, 32 per GPU, UNet 3 -> 6
        h = make_unet(lib, unet_cfg(3, 6, 128, *RES128, F16, 32), lanes, flags)
        mark("sample_iadb B=32 steps=2")
        iadb(lib, dev, h, 32, 3, 3, 128, 2)
    elif name == "c5":                                     # latent cat_res512: UNet 4 -> 8 at B = 8, then the VAE decoder
        h = make_unet(lib, unet_cfg(4, 8, 64, *RES64, F16, 8), lanes, flags)
        mark("sample_iadb B=8 steps=2")
        iadb(lib, dev, h, 8, 4, 4, 64, 2)
        cfg = _lib.VaeConfig()
        cfg.latent_channels, cfg.out_channels, cfg.latent_resolution, cfg.num_levels = 4, 3, 64, 4
        for i, v in enumerate((128, 256, 512, 512)):
            cfg.block_out_channels[i] = v
        cfg.layers_per_block, cfg.dtype, cfg.max_batch = 2, F16, 8
        v = C.c_void_p()
        _lib.check(lib.bndm_vae_decoder_create(C.byref(v), C.byref(cfg)), "vae create")
        load_params(lib, v, 1)
        mark("vae finalize")
        _lib.check(lib.bndm_unet_finalize(v), "vae finalize")
        mark("vae_decode B=8")
        _lib.check(lib.bndm_vae_decode(v, dev.alloc(8 * 4 * 64 * 64 * 4), dev.alloc(8 * 3 * 512 * 512 * 4), 8, None), "vae_decode")
        lib.bndm_unet_destroy(v)
    elif name == "cond":                                   # super-resolution sampler: 6 -> 3 at 128 px, B = 1 (iadb_bn.py:384-438,603)
        h = make_unet(lib, unet_cfg(6, 3, 128, *RES128, F16, 1), lanes, flags)
        mark("sample_iadb conditional B=1 steps=2")
        iadb(lib, dev, h, 1, 3, 6, 128, 2, cond=True)
    elif name == "f32":                                    # fp32-compute verification mode
        h = make_unet(lib, unet_cfg(3, 6, 64, *RES64, F32, 2), 1, 0)
        mark("forward B=2")
        forward(lib, dev, h, 2, 3, 6, 64)
        mark("sample_iadb B=2 steps=2")
        iadb(lib, dev, h, 2, 3, 3, 64, 2)
    elif name == "noise":                                  # bndm_bluenoise + the step kernels, no UNet handle
        L = dev.alloc(4096 * 4096 * 4)
        for (B, Cc, res) in ((2, 3, 64), (64, 3, 64), (2, 3, 128), (32, 3, 128), (8, 4, 64), (2, 4, 32)):
            n = B * Cc * res * res * 4
            z, a, o1, o2, o3 = dev.alloc(n), dev.alloc(B * 4), dev.alloc(n), dev.alloc(n), dev.alloc(n)
            ws = lib.bndm_bluenoise_workspace_bytes(B, Cc, res)
            w = dev.alloc(max(ws, 16))
            for mode in (0, 1):
                mark(f"bluenoise B={B} C={Cc} res={res} mode={mode}")
                _lib.check(lib.bndm_bluenoise(L, 0, z, 0, a, o1, o2, o3, B, 0, B, Cc, res, mode, w, ws, None), "bluenoise")
        mark("bluenoise shard 8..15 of 32 at 128 px")
        n = 32 * 3 * 128 * 128 * 4
        ws = lib.bndm_bluenoise_workspace_bytes(32, 3, 128)
        _lib.check(lib.bndm_bluenoise(L, 0, dev.alloc(n), 0, dev.alloc(128), dev.alloc(n), dev.alloc(n), dev.alloc(n), 32, 8, 8, 3, 128, 0,
                                      dev.alloc(max(ws, 16)), ws, None), "bluenoise shard")
        x, d = dev.alloc(64 * 3 * 4096 * 4), dev.alloc(64 * 6 * 4096 * 4)
        mark("iadb_step / ddim_step / export_u8")
        _lib.check(lib.bndm_iadb_step(x, d, -0.004, -0.004, 64, 3, 6, 4096, None), "iadb_step")
        _lib.check(lib.bndm_iadb_step(x, d, -0.004, 0.0, 64, 3, 3, 4096, None), "iadb_step")
        _lib.check(lib.bndm_ddim_step(x, d, 0.9, 0.43, 0.92, 0.39, 1.0, 64 * 3 * 4096, None), "ddim_step")
        _lib.check(lib.bndm_export_u8(x, dev.alloc(64 * 3 * 4096), 64, 3, 4096, 0, None), "export")
        _lib.check(lib.bndm_export_u8(x, dev.alloc(64 * 3 * 4096), 64, 3, 4096, 1, None), "export")
        return
    else:
        raise SystemExit(f"unknown scenario {name}")
    # the launch list as the ABI reports it (bndm_unet_op_info)
    kern, lab, fl = C.create_string_buffer(128), C.create_string_buffer(256), C.c_double()
    for i in range(lib.bndm_unet_num_ops(h)):
        _lib.check(lib.bndm_unet_op_info(h, i, kern, 128, lab, 256, C.byref(fl)), "op_info")
        mark(f"op {i} {kern.value.decode()} | {' '.join(lab.value.decode().split())} | {fl.value:.0f}")
    mark("destroy")
    lib.bndm_unet_destroy(h)


def main():
    libpath, name = os.path.abspath(sys.argv[1]), sys.argv[2]
    lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    _lib.LIB_PATH = libpath
    if lanes > 1:
        _lib.SIGNATURES["bndm_unet_set_lanes"] = (_lib._i, [_lib._vp, _lib._i, _lib._i])
    lib = _lib.load()
    dev = Dev()
    scenario(lib, dev, name, lanes, flags)
    dev.flush()


if __name__ == "__main__":
    main()
