"""HBM traffic of the fused conv kernel from rocprofv3 PMC passes, recorded for bench.py's roofline.traffic.

Run ON the GPU box from the repo root:   python tools/pmc_traffic.py
Two separate counter passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC
slots") over `python bench.py --profile-only --no-cpu-baseline`, counters only (--kernel-trace --pmc).  Both counters
are in KiB; per the guide's gfx950 note FETCH_SIZE reports half the bytes of a wide coalesced stream, so
    HBM bytes per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.
Writes profiles/pmc_traffic.json {lib_sha256, kernels: {tag: {bytes_per_launch, fetch_kib, write_kib, launches}}};
bench.py reports the figure only while the library's sha256 still matches (never a stale constant)."""
import csv, glob, hashlib, json, os, re, subprocess, sys, tempfile
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "bndm_amd", "libbndm_hip.so")


def family(name):
    m = re.search(r"(conv_t32|conv_s|conv_igemm|gn_small|bluenoise_small|bluenoise_gemm)", name)
    if not m:
        return None
    fam = m.group(1)
    t = re.search(r"Li(16|8)ELi\d+E", name)
    if fam == "conv_t32" and t:
        fam += f"<TH={t.group(1)}>"
    return fam


def one_pass(counter, extra):
    d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--output-format", "csv", "--kernel-trace", "--pmc", counter, "-d", d, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--profile-only", "--no-cpu-baseline"] + extra
    subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                   check=True)
    tot, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            fam = family(r["Kernel_Name"])
            if fam and r["Counter_Name"] == counter:
                tot[fam] += float(r["Counter_Value"])
                n[fam] += 1
    subprocess.run(["rm", "-rf", d])
    return tot, n


def main():
    extra = sys.argv[1:]
    fetch, nf = one_pass("FETCH_SIZE", extra)
    write, nw = one_pass("WRITE_SIZE", extra)
    out = {"lib_sha256": hashlib.sha256(open(LIB, "rb").read()).hexdigest(), "command": "bench.py --profile-only " + " ".join(extra),
           "config": (extra[extra.index("--config") + 1] if "--config" in extra else "c2"),
           "formula": "2 * FETCH_SIZE(KiB) * 1024 + WRITE_SIZE(KiB) * 1024 per launch", "kernels": {}}
    for fam in sorted(fetch):
        if nf[fam] and nw.get(fam):
            fk, wk = fetch[fam] / nf[fam], write[fam] / nw[fam]
            out["kernels"][fam] = {"bytes_per_launch": (2 * fk + wk) * 1024, "fetch_kib": fk, "write_kib": wk, "launches": nf[fam]}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "pmc_traffic.json"), "w") as f:      # gpurun merges gpurun_out/ back
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
