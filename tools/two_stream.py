"""Experiment (not product): the c2 denoising loop as TWO half-batch chains on two HIP streams.

Why: a conv_t32 launch is 2 rounds x (prologue + K loop + epilogue) and the two workgroups of a CU are phase-locked (both in
the prologue, then both in the epilogue: DESIGN section 8), so the MFMA pipe idles for a third of every launch, and every
kernel boundary (94 per forward) drains the chip.  The samples of a batch are independent (GroupNorm and attention are per
sample; tests/test_gpu_benched.py::test_batch_position_independence), so the batch can be cut into two chains of launches
that depend on nothing of each other.  On two streams the dispatcher fills a CU slot freed by chain A's finishing workgroup
with chain B's next kernel: A's epilogue / drain / launch gap is B's K-loop time, without any change to a kernel.

    python tools/two_stream.py [--batch 64] [--nb_steps 250] [--passes 3] [--lanes 2] [lib.so]

Three forms, alternating on one box: the one-stream loop, the chains driven by one host thread each with an engine handle each
(no change to the library at all), and the product form -- bndm_unet_set_lanes: one handle, one host thread, chains enqueued
step by step.

Prints images/s of the one-stream loop (= bench.py's timed region without noise / export) and of the two-stream form on the
same box, alternating, and checks that the results are bit-identical (they must be: same kernels per sample, and the kernel
choice by grid size is the only thing that can differ -- reported if it does)."""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?", default=None)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--nb_steps", type=int, default=250)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--handle-batch", default="full", choices=["full", "lane"],
                    help="max_batch of a chain's engine handle: tile sizes are chosen from it at build time.  'full' keeps the "
                         "B-sample choices (bit-identical results expected), 'lane' lets the engine choose for B / lanes")
    ap.add_argument("--cumask", action="store_true",
                    help="host-thread form on streams with disjoint CU masks (hipExtStreamCreateWithCUMask): chain k gets the "
                         "k-th 32 / lanes bits of every 32-bit mask word, i.e. an equal share of CUs that leaves no XCD empty under "
                         "either bit numbering.  The chains then cannot share a CU, but the chip-wide prologue / epilogue bursts of "
                         "one share run beside the K loops of the other")
    ap.add_argument("--no-stagger", action="store_true", help="all chains start together (default: chain k starts k / lanes of "
                                                              "a forward behind chain 0, so that they do not run in lockstep)")
    ap.add_argument("--forward-ms", type=float, default=3.0, help="host-thread form: forward time assumed for that offset")
    a = ap.parse_args()
    from bndm_amd import _lib
    if a.lib:
        _lib.LIB_PATH = os.path.abspath(a.lib)
    import torch
    from bndm_amd.sampler import get_model, sample_iadb
    from bndm_amd.unet import engine_ops

    dev = torch.device("cuda:0")
    B, N, R, NL = a.batch, a.nb_steps, a.res, a.lanes
    assert B % NL == 0
    params = torch.tensor([1000.0, 0.0, 3.0], device=dev)
    full = get_model(3, 6, R, dtype="f16", seed=0).to(dev).eval()
    lanes = [get_model(3, 6, R, dtype="f16", seed=0).to(dev).eval() for _ in range(NL)]      # same seed = same weights
    def masked_stream(k):
        import ctypes
        path = next((ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln), "libamdhip64.so")
        hip = ctypes.CDLL(path)                                 # the runtime torch itself has loaded
        per = 32 // NL
        word = ((1 << per) - 1) << (k * per)
        words = (ctypes.c_uint32 * 8)(*([word] * 8))            # 256 CUs
        st = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
        if rc != 0:
            raise SystemExit(f"hipExtStreamCreateWithCUMask failed: {rc}")
        return torch.cuda.ExternalStream(st.value, device=dev)

    torch.zeros(1, device=dev)                                  # (runtime loaded and initialised)
    streams = [masked_stream(k) if a.cumask else torch.cuda.Stream(device=dev) for k in range(NL)]
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(B, 3, R, R, generator=g).to(dev)
    hb = B // NL
    if a.handle_batch == "full":
        for m in lanes:
            m._ensure_engine(B, R, dev)          # the handle is kept for any smaller batch

    def one_stream():
        return sample_iadb(full, x0, N, "sigmoid", params, 6, "gaussianBN", "train")

    def multi_stream():
        out = [None] * NL
        cur = torch.cuda.current_stream(dev)
        ev = torch.cuda.Event()
        ev.record(cur)

        def work(i):
            # one host thread per chain: a sampling call enqueues its whole loop (steps x 95 launches) before it returns,
            # and ctypes drops the GIL inside it
            if i and not a.no_stagger:
                time.sleep(i * a.forward_ms * 1e-3 / NL)
            with torch.cuda.device(dev), torch.cuda.stream(streams[i]):
                streams[i].wait_event(ev)
                out[i] = sample_iadb(lanes[i], x0[i * hb:(i + 1) * hb], N, "sigmoid", params, 6, "gaussianBN", "train")
        th = [threading.Thread(target=work, args=(i,)) for i in range(NL)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for s in streams:
            cur.wait_stream(s)
        return torch.cat(out, 0)

    # the in-engine forms exist only in a library built with tools/experiments/lanes.patch (bndm_unet_set_lanes); the product
    # library runs the first two forms, which need no library change, and those decide whether the patch is worth applying
    have_lanes = "bndm_unet_set_lanes" in _lib.SIGNATURES
    if not have_lanes:
        print("(library without bndm_unet_set_lanes: one stream vs engine handle + host thread per chain only)")
    eng = engt = None
    if have_lanes:
        eng = get_model(3, 6, R, dtype="f16", seed=0, lanes=NL, lane_cus=a.cumask, lane_stagger=not a.no_stagger).to(dev).eval()      # chains inside the engine
        engt = get_model(3, 6, R, dtype="f16", seed=0, lanes=NL, lane_threads=True, lane_cus=a.cumask, lane_stagger=not a.no_stagger).to(dev).eval()   # ... one host thread per chain

    def in_engine():
        return sample_iadb(eng, x0, N, "sigmoid", params, 6, "gaussianBN", "train")

    def in_engine_threads():
        return sample_iadb(engt, x0, N, "sigmoid", params, 6, "gaussianBN", "train")

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = fn()
        torch.cuda.synchronize()
        return y, time.perf_counter() - t0

    ya, _ = timed(one_stream)
    yb, _ = timed(multi_stream)
    same = bool(torch.equal(ya, yb))
    if not same:
        d = (ya - yb).abs().max().item()
        print(f"results differ: max abs {d:.3e} (kernel choice by grid size differs between B={B} and B={hb}?)")
        ka = [k for k, _, _ in engine_ops(full._ensure_engine(B, R, dev))]
        kb = [k for k, _, _ in engine_ops(lanes[0]._ensure_engine(hb, R, dev))]
        print("   launch lists differ at ops:", [i for i, (p, q) in enumerate(zip(ka, kb)) if p != q][:20])
    same_c = same_d = None
    if have_lanes:
        yc, _ = timed(in_engine)
        same_c = bool(torch.equal(ya, yc))
        yd, _ = timed(in_engine_threads)
        same_d = bool(torch.equal(ya, yd))
    ta, tb, tc, td = [], [], [], []
    for _ in range(a.passes):
        ta.append(timed(one_stream)[1])
        tb.append(timed(multi_stream)[1])
        if have_lanes:
            tc.append(timed(in_engine)[1])
            td.append(timed(in_engine_threads)[1])
    fa, fb = B / min(ta), B / min(tb)
    if not have_lanes:
        print(f"one stream  B={B}: {fa:8.2f} images/s   ({min(ta) / N * 1e3:.3f} ms per step)   all: {[round(B / t, 1) for t in ta]}")
        print(f"{NL} streams B={hb}x{NL}: {fb:8.2f} images/s   ({min(tb) / N * 1e3:.3f} ms per step)   all: {[round(B / t, 1) for t in tb]}")
        print(f"ratio: engine handle + python thread per chain {fb / fa:.3f} (bit-identical: {same})")
        return
    fc, fd = B / min(tc), B / min(td)
    print(f"one stream  B={B}: {fa:8.2f} images/s   ({min(ta) / N * 1e3:.3f} ms per step)   all: {[round(B / t, 1) for t in ta]}")
    print(f"{NL} streams B={hb}x{NL}: {fb:8.2f} images/s   ({min(tb) / N * 1e3:.3f} ms per step)   all: {[round(B / t, 1) for t in tb]}")
    print(f"in-engine lanes={NL}: {fc:8.2f} images/s   ({min(tc) / N * 1e3:.3f} ms per step)   all: {[round(B / t, 1) for t in tc]}")
    print(f"in-engine lanes={NL}, a host thread per chain: {fd:8.2f} images/s   all: {[round(B / t, 1) for t in td]}")
    print(f"ratio: engine handle + python thread per chain {fb / fa:.3f} (bit-identical: {same})   in-engine {fc / fa:.3f} "
          f"(bit-identical: {same_c})   in-engine + threads {fd / fa:.3f} (bit-identical: {same_d})")


if __name__ == "__main__":
    main()
