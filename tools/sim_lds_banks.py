"""LDS bank conflicts per kernel family from a simulator run with GFX950SIM_STATS=1, next to the hardware counters on record.

The simulator's bank model (tests/gfx950sim/ops.py::_lds_bank_cycles) is the per-instruction banking table of
/opt/skills/guides/MI355X_MICROARCH.md: LDS-array cycles = SQ_LDS_IDX_ACTIVE, extra cycles = SQ_LDS_BANK_CONFLICT.  Conflicts are a function of
addresses alone, so the rates of a batch-1 run are the rates of the benchmark's batch (same per-workgroup work); `profiles/r03_pmc_sq.txt`
(c2, B = 64, driver-run round 3) is the calibration: kernels that are byte-identical to that build must land on its ratios.

    python tools/sim_lds_banks.py <sim_stats.json> [--lib lib.so] [--measured profiles/r03_pmc_sq.txt] [--top N]
"""
import argparse
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def family(sym):
    n = re.sub(r"^_ZN4bndm12_GLOBAL__N_1\d+", "", sym)
    m = re.match(r"conv_t32I(DF16_|DF16b|f)Li(\d+)ELi\dELi(\d)ELi(\d+)", n)
    if m:
        return f"conv_t32<TH={m.group(2)}>" + (" head" if m.group(4) == "32" else ""), f"conv_t32<TH={m.group(2)}>"
    m = re.match(r"conv_s16I", n)
    if m:
        return "conv_s16", None
    m = re.match(r"conv_sI(DF16_|DF16b|f)Li(\d+)ELi(\d)E", n)
    if m:
        return f"conv_s<TM={m.group(2)},NB={m.group(3)}>", "conv_s"
    for k, meas in (("conv_igemm", "conv_igemm"), ("conv_in_kernel", "conv_in"), ("gn_small_kernel", "gn_small"), ("gn_stats_kernel", "gn_stats"),
                    ("temb_mlp_kernel", "temb_mlp"), ("splitk_reduce", "splitk_reduce"), ("head_conv", None), ("bluenoise_gemm", "bluenoise_gemm")):
        if n.startswith(k):
            return k.replace("_kernel", ""), meas
    return re.sub(r"I(DF16_|DF16b|f).*", "", n)[:28], None


def measured(path):
    out, cur = {}, None
    if not path or not os.path.exists(path):
        return out
    for ln in open(path):
        m = re.match(r"^(\S.*?)\s+\((\d+) dispatches\)", ln)
        if m:
            cur = out.setdefault(m.group(1).strip(), {})
            continue
        m = re.match(r"^\s+(SQ_\w+)\s+([\d.e+]+)\s*$", ln)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stats")
    ap.add_argument("--lib", default=None)
    ap.add_argument("--measured", default="profiles/r03_pmc_sq.txt")
    ap.add_argument("--top", type=int, default=3)
    a = ap.parse_args()
    st = json.load(open(a.stats))
    meas = measured(a.measured)
    fam, per_kernel = {}, {}
    for e in st:
        name, mk = family(e["kernel"])
        f = fam.setdefault(name, {"meas": mk, "inst": 0, "cyc": 0, "conf": 0, "launches": 0})
        f["launches"] += 1
        f["inst"] += sum(n for m, n in e["insts"].items() if m.startswith("ds_"))
        f["cyc"] += e["bytes"].get("lds_cycles", 0)
        f["conf"] += e["bytes"].get("lds_conflict", 0)
        pk = per_kernel.setdefault((name, e["kernel"]), {})
        for k, v in e["bytes"].items():
            if k.startswith("ldsc@"):
                pk[k] = pk.get(k, 0) + v
    print(f"{'kernel family':26s} {'launches':>8s} {'ds insts':>9s} {'LDS cycles':>11s} {'conflict':>9s} {'rate':>6s} {'cyc/inst':>8s}   |  measured (B = 64): rate  cyc/inst")
    for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["cyc"]):
        if not f["cyc"]:
            continue
        m = meas.get(f["meas"] or "", {})
        ms = ""
        if m.get("SQ_LDS_IDX_ACTIVE"):
            ms = f"{100 * m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE']:5.1f} %  {m['SQ_LDS_IDX_ACTIVE'] / max(m.get('SQ_INSTS_LDS', 1), 1):5.2f}   ({f['meas']})"
        print(f"{name:26s} {f['launches']:8d} {f['inst']:9d} {f['cyc']:11d} {f['conf']:9d} {100 * f['conf'] / f['cyc']:5.1f}% {f['cyc'] / max(f['inst'], 1):8.2f}   |  {ms}")
    if a.top:
        from tests.gfx950sim import loader
        ks = loader.load_library(a.lib or "bndm_amd/libbndm_hip.so")
        print("\nwhere the conflict cycles are (instruction address: cycles):")
        for (name, sym), pk in sorted(per_kernel.items()):
            tot = sum(pk.values())
            if not tot:
                continue
            by = {i.addr: i for i in ks[sym].insts}
            # fold unrolled copies: same text modulo register numbers
            folded = {}
            for k, v in pk.items():
                t = re.sub(r"v\[\d+:\d+\]|v\d+", "v", by[int(k[5:], 16)].text)
                folded[t] = folded.get(t, 0) + v
            print(f"  {name}  [{re.sub(r'^_ZN4bndm12_GLOBAL__N_1[0-9]+', '', sym)[:40]}]  {tot} conflict cycles")
            for t, v in sorted(folded.items(), key=lambda kv: -kv[1])[:a.top]:
                print(f"      {100 * v / tot:5.1f} %  {t}")


if __name__ == "__main__":
    main()
