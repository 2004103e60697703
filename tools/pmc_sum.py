"""Summarise rocprofv3 --pmc output directories: per kernel family, the sum of every counter over all dispatches.
usage: python tools/pmc_sum.py DIR [DIR...] [--match substr]"""
import csv, glob, os, re, sys
from collections import defaultdict

dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
match = None
if "--match" in sys.argv:
    match = sys.argv[sys.argv.index("--match") + 1]
    dirs = [d for d in dirs if d != match]


def family(name):
    m = re.search(r"(conv_t32|conv_s|conv_tap9s|conv_tap9|conv_fused|conv_igemm|conv_lowres|gn_small|gn_finalize2|gn_stats|gn_apply|"
                  r"attention|bluenoise_gemm|bluenoise_finish|conv_in|splitk_reduce|temb_mlp|conv_out|attn_block)", name)
    fam = m.group(1) if m else name[:40]
    t = re.search(r"Li(16|8)ELi\d+E", name)
    if fam in ("conv_t32", "conv_tap9") and t:
        fam += f"<TH={t.group(1)}>"
    return fam


tot = defaultdict(lambda: defaultdict(float))
ndisp = defaultdict(set)
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            fam = family(row["Kernel_Name"])
            if match and match not in fam:
                continue
            tot[fam][row["Counter_Name"]] += float(row["Counter_Value"])
            ndisp[fam].add((f, row["Dispatch_Id"]))
for fam in sorted(tot):
    c = tot[fam]
    print(f"{fam}  ({len(ndisp[fam])} dispatches)")
    for k in sorted(c):
        print(f"    {k:32s} {c[k]:.4g}")
    if "SQ_WAVE_CYCLES" in c:
        wc = c["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in c:
                print(f"    {k}/WAVE_CYCLES = {c[k] / wc:.3f}")
    if "SQ_INSTS_MFMA" in c and c["SQ_INSTS_MFMA"]:
        m = c["SQ_INSTS_MFMA"]
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM"):
            if k in c:
                print(f"    {k}/MFMA = {c[k] / m:.2f}")
    if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
        print(f"    LDS conflict share = {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.3f}")
