#!/bin/bash
# profiling aid: sum of the per-op times of the low-resolution (unfused) convolutions under env knob settings
# usage: tools/lowres.sh "ENV1=a ENV2=b" "ENV1=c" ...
mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg BNDM_PROFILE_DUMP=gpurun_out/lowres_$i.txt python bench.py --profile-only --no-cpu-baseline > gpurun_out/lowres_$i.log 2>&1
  python - "$cfg" gpurun_out/lowres_$i.txt <<'PY'
import sys,re,collections
tot=collections.defaultdict(float)
for ln in open(sys.argv[2]):
    p=ln.split()
    if len(p)<6 or p[2]!='ms': continue
    m=re.search(r'(\d+)x(\d+)\s*$',ln)
    if p[5]=='conv' and m and int(m.group(1))<=8: tot[m.group(1)]+=float(p[1])
    tot['all']+=float(p[1])
print(sys.argv[1], {k: round(v,3) for k,v in tot.items()})
PY
done
