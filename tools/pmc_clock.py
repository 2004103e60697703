"""Effective shader clock per kernel family from a rocprofv3 pass with --kernel-trace --pmc GRBM_GUI_ACTIVE:
sum of the counter over the family's dispatches / sum of their durations.  (If the counter comes back summed over the 8 XCDs the
figure is 8x a plausible clock: both readings are printed.)      python tools/pmc_clock.py DIR"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def family(n):
    m = re.search(r"(conv_t32|conv_s|conv_igemm|gn_small|gn_stats|conv_in|splitk_reduce|iadb_step|bluenoise_\w+)", n)
    fam = m.group(1) if m else n[:40]
    t = re.search(r"Li(32|16|8)ELi\d+ELi(4|8)E", n)
    return fam + (f"<TH={t.group(1)},{t.group(2)} waves>" if fam == "conv_t32" and t else "")


d = sys.argv[1]
dur = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), family(r["Kernel_Name"]))
act = defaultdict(float)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            act[r["Dispatch_Id"]] += float(r["Counter_Value"])
tot = defaultdict(lambda: [0.0, 0.0, 0])
for k, v in act.items():
    if k in dur:
        t = tot[dur[k][1]]
        t[0] += v
        t[1] += dur[k][0]
        t[2] += 1
print("effective clock (GRBM_GUI_ACTIVE / duration), GHz:  as counted | / 8 XCDs")
for fam, (a, ns, n) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"    {fam:34s} {n:6d} dispatches  {a / ns:7.3f} | {a / ns / 8:6.3f}")
