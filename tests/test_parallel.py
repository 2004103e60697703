"""world_size-2 gloo tests of the batch-sharding helpers (the N>1 path of bench.py / the CLI)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bndm_amd.parallel import gather_images, max_over_ranks, shard_range


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 64, 65, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0
            assert sum(c for _, c in spans) == total
            for (b0, c0), (b1, _) in zip(spans, spans[1:]):
                assert b0 + c0 == b1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = (torch.arange(total * 2 * 2 * 3, dtype=torch.int64) % 251).to(torch.uint8).view(total, 2, 2, 3)
        b0, bc = shard_range(total, rank, world)
        counts = [shard_range(total, r, world)[1] for r in range(world)]
        got = gather_images(full[b0:b0 + bc].clone(), counts, dst=0)
        mx = max_over_ranks(float(rank + 1), device="cpu")
        ok = (mx == world) and ((got is None) if rank else torch.equal(got, full))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_gather_images_world2_gloo(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_bench_self_launch_world2_dry_run():
    """`python bench.py --gpus 2` with no launcher starts its own ranks: dry run on CPU / gloo --
    two ranks rendezvous on 127.0.0.1, barrier, max-over-ranks, ONE JSON line from rank 0 carrying n_gpus = 2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-cpu"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["dry_run"] is True
    assert lines[0]["elapsed"] >= 0.02                                   # the slower rank's time (max over ranks)


def test_bench_refuses_a_mislabelled_world():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-cpu"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "refusing" in (out.stderr + out.stdout)
