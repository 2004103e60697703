#!/bin/bash
# Step 4 of tools/next_gpu_session.sh: is cutting the batch into independent chains of launches inside one GPU worth anything?
# Part 1 needs no library change.  Part 2 applies tools/experiments/lanes.patch to THIS COPY of the tree (the GPU box's scratch
# copy; never on the build box) and swaps in the prebuilt tools/lib_lanes.so.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; exec > >(tee gpurun_out/r05_lanes.log) 2>&1
echo "== 1. product library: one stream vs one engine handle + host thread per chain (2 and 4 chains; shared CUs, then CU shares)"
timeout 600 python tools/two_stream.py --passes 2 --nb_steps 100 --lanes 2 2>&1 | tail -4
timeout 600 python tools/two_stream.py --passes 2 --nb_steps 100 --lanes 4 2>&1 | tail -4
timeout 600 python tools/two_stream.py --passes 2 --nb_steps 100 --lanes 4 --cumask 2>&1 | tail -4
echo "== 1a. which CUs does a CU-masked stream really get?  (tools/ubench/cumask_probe: XCC_ID / CU_ID histogram per mask)"
[ -x tools/ubench/cumask_probe.bin ] && timeout 60 tools/ubench/cumask_probe.bin
echo "== 2. lanes.patch on this scratch copy"
if [ "$R" = /root/repo ] && [ -d /root/repo/.git ]; then echo "refusing to patch the build box's tree"; exit 1; fi
git apply tools/experiments/lanes.patch && cp tools/lib_lanes.so bndm_amd/libbndm_hip.so || exit 1
timeout 900 python -m pytest tools/experiments/extra_tests/test_gpu_lanes.py tools/experiments/extra_tests/test_lanes_plumbing.py -m "gpu or not gpu" -q -x 2>&1 | tail -5
for n in "1" "2" "4" "4 --lane-cus" "4 --lane-threads"; do
  echo -n "-- bench.py --lanes $n:  "
  timeout 600 python bench.py --lanes $n --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'images/s', d['ms_per_step'], 'ms per pass', 'end_to_end_frac', d['roofline'].get('end_to_end_frac'))"
done
echo "-- c5 (B = 8 -> 2 x 4) with lanes"
timeout 600 python bench.py --config c5 --lanes 2 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
echo "-- kernel trace of a 4-lane run: share of the time with kernels of more than one queue in flight"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace -d $R/gpurun_out/lanes_kt -- python $R/bench.py --lanes 4 --steps 1 --warmup 1 --nb_steps 25 --no-cpu-baseline --no-other-configs > $R/gpurun_out/lanes_kt.log 2>&1)
python tools/overlap.py gpurun_out/lanes_kt 2>&1 | tee gpurun_out/r05_lanes_overlap.txt | head -20; rm -rf gpurun_out/lanes_kt
