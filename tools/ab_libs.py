"""Profiling aid: same-box A/B of several builds of the library (interleaved rounds, one subprocess per run).

    python tools/ab_libs.py [--rounds 3] [--acc] [--full] libA.so libB.so libB.so@BNDM_TH32_MIN=256 ...

Per library and round: `bench.py --profile-only` (HIP-event per-op profile of the c2 forward, B=64) -> ms per forward, the
average conv_t32<TH=16> launch, per-family / per-resolution sums.  --acc adds the rel-L2 of the f16 engine against the fp32
HIP mode at B=64 (the 2e-3 bar of tests/test_gpu_benched.py); --full adds a 2-pass timed run (images/s)."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def split_variant(v):
    """'path/lib.so@VAR=val,VAR2=val2' -> (path, {VAR: val, ...}): one library under different environment switches"""
    lib, _, envs = v.partition("@")
    return os.path.abspath(lib), dict(kv.split("=", 1) for kv in envs.split(",") if kv)


def run(variant, extra, dump=None):
    lib, extra_env = split_variant(variant)
    env = dict(os.environ, **extra_env)
    if dump:
        env["BNDM_PROFILE_DUMP"] = dump
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_with_lib.py"), lib, "--no-cpu-baseline",
                        "--no-other-configs"] + extra, capture_output=True, text=True, env=env, cwd=ROOT)
    for ln in reversed(r.stdout.strip().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    raise SystemExit(f"{lib}: no JSON line\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")


def families(dump):
    fam = collections.defaultdict(float)
    for ln in open(dump):
        m = re.match(r"\s*\d+\s+([\d.]+) ms\s+[\d.]+ TF/s\s+(\S+)\s+(.*)", ln)
        if not m:
            continue
        t, kind, rest = float(m.group(1)), m.group(2), m.group(3).split()
        hw = rest[-1] if rest and "x" in rest[-1] else "-"
        fam[f"{kind}@{hw}"] += t
    return fam


ACC = r"""
import sys, torch
sys.path.insert(0, %r)
from bndm_amd import _lib
_lib.LIB_PATH = %r
from bndm_amd.sampler import get_model
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(64, 3, 64, 64, generator=g).to(dev)
t = torch.linspace(0.004, 1.0, 64).to(dev)
m16 = get_model(3, 6, 64, dtype="f16", seed=0).to(dev).eval()
m32 = get_model(3, 6, 64, dtype="f32", seed=0).to(dev).eval()
with torch.no_grad():
    a = m16(x, t, return_dict=False)[0].double()
    b = torch.cat([m32(x[i:i + 8], t[i:i + 8], return_dict=False)[0] for i in range(0, 64, 8)]).double()   # (fp32 mode: max_batch 8)
rel = float((a - b).norm() / b.norm())
worst = max(float((a[i] - b[i]).norm() / b[i].norm()) for i in range(64))
print("ACC %%.4e %%.4e" %% (rel, worst))
"""


def main():
    args = sys.argv[1:]
    rounds, acc, full = 3, False, False
    libs = []
    while args:
        a = args.pop(0)
        if a == "--rounds":
            rounds = int(args.pop(0))
        elif a == "--acc":
            acc = True
        elif a == "--full":
            full = True
        else:
            libs.append(a)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    res = {l: [] for l in libs}
    for r in range(rounds):
        for i, lib in enumerate(libs):
            dump = os.path.join(ROOT, "gpurun_out", f"ab_{i}_{r}.txt")
            d = run(lib, ["--profile-only"], dump)
            rf = d["roofline"]
            rec = {"fwd": rf["ms_per_forward_total"], "t16": rf["avg_launch_us"], "fam": families(dump)}
            if full:
                rec["ips"] = run(lib, ["--steps", "2", "--warmup", "1"])["value"]
            res[lib].append(rec)
            print(f"round {r} {os.path.basename(lib)}: fwd {rec['fwd']:.3f} ms  t32<16> {rec['t16']:.2f} us" +
                  (f"  {rec['ips']:.2f} img/s" if full else ""), flush=True)
    for lib in libs:
        rs = res[lib]
        best = min(rs, key=lambda x: x["fwd"])
        med = sorted(x["fwd"] for x in rs)[len(rs) // 2]
        fam = collections.defaultdict(float)
        for x in rs:
            for k, v in x["fam"].items():
                fam[k] += v / len(rs)
        top = sorted(fam.items(), key=lambda kv: -kv[1])[:12]
        print(f"== {os.path.basename(lib)}: fwd min {best['fwd']:.3f} med {med:.3f} ms; t32<16> min "
              f"{min(x['t16'] for x in rs):.2f} us" + (f"; img/s max {max(x['ips'] for x in rs):.2f}" if full else ""))
        print("   " + "  ".join(f"{k} {v:.3f}" for k, v in top))
    if acc:
        for lib in libs:
            lp, ev = split_variant(lib)
            r = subprocess.run([sys.executable, "-c", ACC % (ROOT, lp)], capture_output=True, text=True, cwd=ROOT,
                               env=dict(os.environ, **ev))
            print(f"== {os.path.basename(lib)}: " + (r.stdout.strip().splitlines() or [r.stderr[-500:]])[-1])


if __name__ == "__main__":
    main()
