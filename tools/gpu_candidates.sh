#!/bin/bash
# Step 3 of tools/next_gpu_session.sh: drain the candidate queue, singles only.  Run from the repo root on the GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; exec > >(tee gpurun_out/r05_candidates.log) 2>&1
echo "shipped: $(sha256sum bndm_amd/libbndm_hip.so)"
echo "== hashes (c2, c4): the three bit-identical-by-construction candidates must print the shipped library's line"
for c in c2 c4; do for l in bndm_amd/libbndm_hip.so tools/lib_v9.so tools/lib_v12.so tools/lib_v13.so; do
  echo -n "$c $l  "; timeout 300 python tools/fwd_hash.py $l $c 2>&1 | tail -1; done; done
echo "== A/B, two interleaved rounds: shipped | v9 staged 1x1 chunks | v12 scalar chunk descriptors | v13 first-round write-back"
timeout 1500 python tools/ab_libs.py --rounds 2 --acc bndm_amd/libbndm_hip.so tools/lib_v9.so tools/lib_v12.so tools/lib_v13.so 2>&1 | tail -16
echo "== v8 (round-4 patch: pair-granular sums, sums-first prologue; conv_t32<TH=32> only with BNDM_TH32_MIN): accuracy + A/B"
timeout 1200 python tools/ab_libs.py --rounds 2 --acc bndm_amd/libbndm_hip.so tools/lib_v8.so "tools/lib_v8.so@BNDM_TH32_MIN=256" 2>&1 | tail -12
echo "== first-level widths 64 / 256 (parked test; on the shipped library)"
timeout 600 python -m pytest tools/experiments/extra_tests/test_gpu_first_level_widths.py -m gpu -q 2>&1 | tail -4
