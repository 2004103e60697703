"""Dynamic instruction mix per kernel family from a simulator run with GFX950SIM_STATS=1 (tests/hipmock/exec_forward.py writes
sim_stats_<case>.json next to its outputs): executed wave-instructions by class, and the bytes its vector memory instructions moved.
    python tools/sim_instruction_mix.py <sim_stats.json> [kernel substring]"""
import json
import re
import sys


def cls_of(m):
    if m.startswith("v_mfma"):
        return "mfma"
    if m.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_sin", "v_cos")):
        return "trans"
    if m.startswith("v_"):
        return "valu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("buffer_", "global_")):
        return "vmem"
    if m in ("s_waitcnt", "s_barrier", "s_nop"):
        return "wait"
    return "salu"


def main():
    st = json.load(open(sys.argv[1]))
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    print(f"{'kernel':44s} {'grid':>12s} {'total':>9s} {'mfma':>8s} {'valu':>8s} {'trans':>7s} {'salu':>8s} {'lds':>7s} {'vmem':>6s} {'wait':>7s} {'loadMB':>7s} {'storeMB':>7s} {'LDScyc':>9s} {'conflict':>8s}")
    for e in st:
        if sub not in e["kernel"]:
            continue
        c = {}
        for m, n in e["insts"].items():
            c[cls_of(m)] = c.get(cls_of(m), 0) + n
        name = re.sub(r"^_ZN4bndm12_GLOBAL__N_1\d+", "", e["kernel"])[:44]
        g = "x".join(str(x) for x in e["grid"])
        print(f"{name:44s} {g:>12s} {sum(c.values()):9d} {c.get('mfma', 0):8d} {c.get('valu', 0):8d} {c.get('trans', 0):7d} {c.get('salu', 0):8d} "
              f"{c.get('lds', 0):7d} {c.get('vmem', 0):6d} {c.get('wait', 0):7d} {e['bytes'].get('load', 0) / 1e6:7.2f} {e['bytes'].get('store', 0) / 1e6:7.2f} "
              f"{e['bytes'].get('lds_cycles', 0):9d} {100.0 * e['bytes'].get('lds_conflict', 0) / max(e['bytes'].get('lds_cycles', 0), 1):7.1f}%")


if __name__ == "__main__":
    main()
