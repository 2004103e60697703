#!/bin/bash
# profiling aid: A/B of one environment switch.  usage: tools/ab_env.sh VAR val1 val2 ...   -> per-family sums of the per-op HIP-event profile
var=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  env $var=$v BNDM_PROFILE_DUMP=gpurun_out/ab_${var}_$v.txt python bench.py --profile-only --no-cpu-baseline > gpurun_out/ab_${var}_$v.json 2>gpurun_out/ab_${var}_$v.err
  python - <<PY
import re,collections
fam=collections.defaultdict(lambda:[0,0.0])
tot=0
for ln in open("gpurun_out/ab_${var}_$v.txt"):
    m=re.match(r"\s*\d+\s+([\d.]+) ms\s+[\d.]+ TF/s\s+(\S+)",ln)
    if not m: continue
    t=float(m.group(1)); k=m.group(2); fam[k][0]+=1; fam[k][1]+=t; tot+=t
print("$var=$v total %.3f ms  "%tot + "  ".join("%s %d/%.3f"%(k,n,t) for k,(n,t) in sorted(fam.items(), key=lambda x:-x[1][1])))
PY
done
