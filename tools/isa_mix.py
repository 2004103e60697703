"""Static instruction mix of the loops of a kernel that hold at least N MFMAs, from device assembly
(hipcc --offload-arch=gfx950 -O3 -std=c++17 [-fno-slp-vectorize] -I include --cuda-device-only -S file.hip -o file.s):

    python tools/isa_mix.py file.s <substring of the mangled kernel name> [min MFMAs per loop = 100]

A loop = a label and the last branch back to it.  Works without a GPU: the figure the round-3 review asked for (VALU per MFMA) as the
compiler emitted it, per loop, next to the SQ counters of profiles/ (which also count the prologue and the epilogue)."""
import collections
import re
import sys

src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
nmin = int(sys.argv[3]) if len(sys.argv) > 3 else 100
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and key in l and ":" in l)
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
k = src[start:end]
print(k[0].split(":")[0])
labels = {m.group(1): i for i, l in enumerate(k) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
loops = {}
for i, l in enumerate(k):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops[labels[m.group(1)]] = i
for a, b in sorted(loops.items()):
    c = collections.Counter()
    for l in k[a:b + 1]:
        t = l.strip().split()
        if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
            continue
        op = t[0]
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            c["trans"] += bool(re.match(r"v_(exp|rcp|log|sqrt|rsq|sin|cos)", op))
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("buffer_", "global_")):
            c["vmem"] += 1
        elif op.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif op.startswith("s_barrier"):
            c["barrier"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    if c["mfma"] >= nmin:
        n = c["mfma"]
        print(f"  loop at lines {a}..{b}: {n} MFMA | VALU {c['valu']} ({c['valu'] / n:.2f} per MFMA, {c['trans']} transcendental) | "
              f"LDS {c['lds']} ({c['lds'] / n:.2f}) | SALU {c['salu']} ({c['salu'] / n:.2f}) | VMEM {c['vmem']} | waits {c['waitcnt']} | barriers {c['barrier']}")
