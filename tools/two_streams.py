"""Experiment: two half-batches on two streams/handles (low-res layers of one overlap hi-res layers of the other)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bndm_amd.sampler import get_model, sample_iadb
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
params = torch.tensor([1000.0, 0.0, 3.0], device=dev)
def run(model, x, stream, out, i):
    with torch.cuda.stream(stream):
        out[i] = sample_iadb(model, x, N, "sigmoid", params, 6, "gaussianBN", "train")
for nstreams in (1, 2, 4):
    Bs = 64 // nstreams
    models = [get_model(3, 6, 64, seed=0).to(dev) for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    xs = [torch.randn(Bs, 3, 64, 64, device=dev) for _ in range(nstreams)]
    out = [None] * nstreams
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(models[i], xs[i], streams[i], out, i)) for i in range(nstreams)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"streams={nstreams} B/stream={Bs}: {dt / N * 1e3:.3f} ms per step for 64 images -> {64 / (dt / N * 250):.1f} img/s at 250 steps", flush=True)
    for m in models: m.release_engine()
