"""How much of a run has kernels of more than one queue in flight -- from rocprofv3 --kernel-trace CSV output.

    rocprofv3 --kernel-trace -d gpurun_out/lanes_trace -- python bench.py --lanes 4 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs
    python tools/overlap.py gpurun_out/lanes_trace

Prints, over the span of the trace: time with 0 / 1 / 2 / 3+ kernels in flight, the same split by number of distinct queues, and
per kernel family the mean duration when running alone vs beside a kernel of another queue (what lanes cost a kernel)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def family(n):
    m = re.search(r"(conv_t32|conv_s|conv_igemm|gn_small|gn_stats|conv_in|splitk_reduce|iadb_step|ddim_step|bluenoise_\w+|export_u8)", n)
    return m.group(1) if m else n[:40]


rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), family(r["Kernel_Name"])))
if not rows:
    raise SystemExit("no *kernel_trace.csv under " + sys.argv[1])
rows.sort()
ev = []
for i, (s, e, q, _) in enumerate(rows):
    ev.append((s, 1, i))
    ev.append((e, -1, i))
ev.sort()
live, by_n, by_q, last = set(), defaultdict(int), defaultdict(int), ev[0][0]
shared = [0] * len(rows)          # ns during which kernel i ran beside a kernel of another queue
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        by_n[min(len(live), 3)] += dt
        qs = {rows[j][2] for j in live}
        by_q[min(len(qs), 3)] += dt
        if len(qs) > 1:
            for j in live:
                shared[j] += dt
    last = t
    (live.add if d > 0 else live.discard)(i)
span = ev[-1][0] - ev[0][0]
print(f"{len(rows)} kernels over {span / 1e6:.3f} ms, queues: {sorted({r[2] for r in rows})}")
print("kernels in flight :", "  ".join(f"{k}{'+' if k == 3 else ''}: {100 * v / span:5.1f}%" for k, v in sorted(by_n.items())))
print("distinct queues   :", "  ".join(f"{k}{'+' if k == 3 else ''}: {100 * v / span:5.1f}%" for k, v in sorted(by_q.items())))
alone, beside = defaultdict(list), defaultdict(list)
for (s, e, q, fam), sh in zip(rows, shared):
    (beside if sh > 0.5 * (e - s) else alone)[fam].append((e - s) / 1e3)
print(f"{'family':20s} {'alone n':>8s} {'avg us':>8s} {'beside n':>9s} {'avg us':>8s}")
for fam in sorted(set(alone) | set(beside), key=lambda k: -(sum(alone[k]) + sum(beside[k]))):
    a, b = alone[fam], beside[fam]
    print(f"{fam:20s} {len(a):8d} {sum(a) / len(a) if a else 0:8.2f} {len(b):9d} {sum(b) / len(b) if b else 0:8.2f}")
