"""Per-kernel-family duration statistics from rocprofv3 --kernel-trace CSV output (*_kernel_trace.csv)."""
import csv, glob, os, re, sys
from collections import defaultdict
d = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"(conv_t32|conv_s|conv_igemm|gn_small|gn_finalize2|gn_stats|gn_apply|attention|bluenoise_small|bluenoise_gemm|"
                      r"bluenoise_finish|conv_in|splitk_reduce|temb_mlp|iadb_step|export_u8|softmax_rows|conv_out|conv_f32|gn_f32)", n)
        fam = m.group(1) if m else n[:50]
        t = re.search(r"Li(16|8)ELi\d+E", n)
        if fam == "conv_t32" and t:
            fam += f"<TH={t.group(1)}>"
        ts = re.search(r"conv_sIDF16b?_?Li(\d+)ELi(\d+)E", n)
        if fam == "conv_s" and ts:
            fam += "<qkv+attention>" if ts.group(2) == "3" else f"<TM={ts.group(1)}>"
        if fam == "bluenoise_small":
            fam += "<16x16x4>" if "Lb1" in n else "<32x32x2>"
        d[fam].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
print(f"{'kernel':34s} {'calls':>7s} {'total us':>12s} {'avg us':>9s} {'min us':>9s} {'share':>7s}")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:34s} {len(v):7d} {sum(v):12.1f} {sum(v)/len(v):9.2f} {min(v):9.2f} {100*sum(v)/tot:6.1f}%")
