#!/bin/bash
# What to run first when a GPU is available again (round 4 ended with GPU access closed; see profiles/r04_README.md).
# Candidate libraries are built from the patches under tools/experiments/ (tools/build_candidates.sh); everything is A/B'd on one
# box against the shipped library before anything is applied to the product sources.
#   on the build box first:   make -C bndm_amd/csrc && bash tools/build_candidates.sh && bash tools/ubench/build.sh
#   usage (on the GPU box, from the repo root):  bash tools/next_gpu_session.sh 2>&1 | tee gpurun_out/next_session.log
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; mkdir -p gpurun_out
echo "== 1. smoke of the shipped library"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== 2. lanes (bndm_unet_set_lanes): parked tests, then one stream vs host-thread chains vs in-engine chains, 2 and 4 lanes"
python -m pytest tools/experiments/extra_tests/test_gpu_lanes.py -m gpu -q -x 2>&1 | tail -5
python tools/two_stream.py --passes 2 --nb_steps 100 2>&1 | tail -5
python tools/two_stream.py --passes 2 --nb_steps 100 --lanes 4 2>&1 | tail -5
python tools/two_stream.py --passes 2 --nb_steps 100 --lanes 4 --no-stagger 2>&1 | tail -3
echo "-- chains on disjoint CU shares (host-thread form only; the line to read is the 'streams' one)"
timeout 600 python tools/two_stream.py --passes 2 --nb_steps 100 --lanes 2 --cumask 2>&1 | tail -5
timeout 600 python tools/two_stream.py --passes 2 --nb_steps 100 --lanes 4 --cumask 2>&1 | tail -5
for n in "1" "2" "4" "2 --lane-cus" "4 --lane-cus" "4 --lane-threads"; do
  echo "-- bench.py --lanes $n"
  python bench.py --lanes $n --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   c2', d['value'], 'images/s', d['ms_per_step'], 'ms per pass')"
done
echo "-- (other configurations under the best arrangement: python bench.py --lanes N [--lane-cus] --steps 2 --warmup 1 --no-cpu-baseline)"
echo "-- kernel trace of a 4-lane run: how much of the time kernels of more than one queue are in flight"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --output-format csv --kernel-trace -d $R/gpurun_out/lanes_kt -- python $R/bench.py --lanes 4 --steps 1 --warmup 1 --nb_steps 25 --no-cpu-baseline --no-other-configs > $R/gpurun_out/lanes_kt.log 2>&1)
python tools/overlap.py gpurun_out/lanes_kt 2>&1 | tee gpurun_out/lanes_overlap.txt | head -20; rm -rf gpurun_out/lanes_kt
echo "== 3. staged 1x1 (shortcut) chunks (v9) and scalar chunk descriptors on the product sources (v12): must hash like the shipped library"
for c in c2 c4; do for l in bndm_amd/libbndm_hip.so tools/lib_v9.so tools/lib_v12.so tools/lib_v13.so tools/lib_v14.so; do echo -n "$l  "; python tools/fwd_hash.py $l $c 2>&1 | tail -1; done; done
echo "== 3a. A/B: shipped vs candidates (per-op profile, accuracy vs the fp32 mode)"
python tools/ab_libs.py --rounds 2 --acc bndm_amd/libbndm_hip.so tools/lib_v9.so tools/lib_v12.so tools/lib_v13.so tools/lib_v14.so tools/lib_v6.so tools/lib_v7.so tools/lib_v11.so "tools/lib_v8.so@BNDM_TH32_MIN=256" 2>&1 | tail -20
echo "== 4. grid sweep of single conv_t32 launches: shipped TH=16, candidate TH=16 and TH=32"
tools/ubench/t32_bench.bin 16 1 0
mkdir -p /tmp/cand && cp tools/lib_v8.so /tmp/cand/libbndm_hip.so
LD_LIBRARY_PATH=/tmp/cand tools/ubench/t32_bench.bin 16 1 0
LD_LIBRARY_PATH=/tmp/cand tools/ubench/t32_bench.bin 32 1 0
echo "== 4a. can launches of one stream overlap (hipExtAnyOrderLaunch)?"; timeout 60 tools/ubench/anyorder.bin
echo "== 4a2. producer -> consumer hand-over inside one launch (bounded spin): which load form is coherent across XCDs, and the latency"; timeout 60 tools/ubench/flagwait.bin
echo "== 4b. never-run extra tests (first-level widths 64 / 256)"
python -m pytest tools/experiments/extra_tests/test_gpu_first_level_widths.py -m gpu -q 2>&1 | tail -5
echo "== 5. if a candidate wins: apply its patch, rebuild, then the full suite:  python -m pytest tests -m gpu -x -q"
echo "== 6. capture: bash tools/profile_round.sh r04   (copies to profiles/ by hand)"
