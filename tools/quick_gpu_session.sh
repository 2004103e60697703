#!/bin/bash
# The 12-minute version of tools/next_gpu_session.sh for a short GPU window: smoke, the parked lanes tests, one lanes A/B, the two
# bit-identical kernel candidates' hashes and one A/B round.    bash tools/quick_gpu_session.sh 2>&1 | tee gpurun_out/quick_session.log
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; mkdir -p gpurun_out
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== lanes tests"; timeout 900 python -m pytest tools/experiments/extra_tests/test_gpu_lanes.py -m gpu -q -x 2>&1 | tail -4
echo "== lanes: one stream vs chains (4 lanes; shared CUs, then CU shares)"
timeout 600 python tools/two_stream.py --passes 2 --nb_steps 60 --lanes 4 2>&1 | tail -5
timeout 600 python tools/two_stream.py --passes 2 --nb_steps 60 --lanes 4 --cumask 2>&1 | tail -5
echo "== hashes: shipped / staged 1x1 chunks / scalar chunk descriptors"
for l in bndm_amd/libbndm_hip.so tools/lib_v9.so tools/lib_v12.so tools/lib_v13.so tools/lib_v14.so; do echo -n "$l  "; timeout 300 python tools/fwd_hash.py $l c2 2>&1 | tail -1; done
echo "== A/B (one round)"; timeout 900 python tools/ab_libs.py --rounds 1 bndm_amd/libbndm_hip.so tools/lib_v9.so tools/lib_v12.so tools/lib_v13.so tools/lib_v14.so tools/lib_v11.so 2>&1 | tail -10
