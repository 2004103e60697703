#!/bin/bash
# Step 3 of tools/next_gpu_session.sh: drain the candidate queue, singles only.  Run from the repo root on the GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; exec > >(tee gpurun_out/r06_candidates.log) 2>&1
echo "shipped: $(sha256sum bndm_amd/libbndm_hip.so)"
echo "== hashes (c2, c4): the three bit-identical-by-construction candidates (v9, v12, v19) must print the shipped library's line"
for c in c2 c4; do for l in bndm_amd/libbndm_hip.so tools/lib_v9.so tools/lib_v12.so tools/lib_v19.so; do
  echo -n "$c $l  "; timeout 300 python tools/fwd_hash.py $l $c 2>&1 | tail -1; done; done
echo "== A/B, two interleaved rounds: shipped | v9 staged 1x1 chunks | v12 scalar chunk descriptors | v19 conv_s bank-conflict-free LDS"
timeout 1500 python tools/ab_libs.py --rounds 2 --acc bndm_amd/libbndm_hip.so tools/lib_v9.so tools/lib_v12.so tools/lib_v19.so 2>&1 | tail -16
echo "== v19's prediction (profiles/r06_sim_lds_banks.txt): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of conv_s 20 % on the shipped library, ~1 % (q|k|v launches only) on v19"
for l in bndm_amd/libbndm_hip.so tools/lib_v19.so; do
  d=/tmp/pmc_lds_$(basename $l .so); rm -rf $d
  (cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS -d $d -- python $R/tools/fwd_hash.py $R/$l c2 > /dev/null 2>&1)
  echo "-- $l"; python tools/pmc_sum.py $d --match conv_s 2>&1 | tail -6
done
echo "== v8 (round-4 patch: pair-granular sums, sums-first prologue; conv_t32<TH=32> only with BNDM_TH32_MIN): accuracy + A/B"
timeout 1200 python tools/ab_libs.py --rounds 2 --acc bndm_amd/libbndm_hip.so tools/lib_v8.so "tools/lib_v8.so@BNDM_TH32_MIN=256" 2>&1 | tail -12
echo "== c5 (latent UNet, B = 8 per GPU) under the EXISTING switches: is a batch-size heuristic between tested code paths worth anything?"
for e in "" "BNDM_NO_TAIL=1" "BNDM_TH16_MIN=1"; do
  echo -n "-- c5 ${e:-default}:  "
  env $e timeout 600 python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(d['value'], 'images/s;', r.get('ms_per_forward_total'), 'ms per forward,', r.get('launches_total'), 'launches')"
done
echo "== v16 (64-channel conv_t32 n-tiles for small-batch handles): c5 with BNDM_NCO64_MAX = 128 / 256 / 512 against the shipped library"
mkdir -p /tmp/v16 && cp -r bndm_amd tests oracle include bluenoise utils.py iadb_bn.py ddim_diffusers.py latent_iadb_bn_diffusers.py input_args.py pytest.ini __graft_entry__.py bench.py /tmp/v16/ 2>/dev/null && cp tools/lib_v16.so /tmp/v16/bndm_amd/libbndm_hip.so
for e in "" "BNDM_NCO64_MAX=128" "BNDM_NCO64_MAX=256" "BNDM_NCO64_MAX=512"; do
  echo -n "-- c5 v16 ${e:-off}:  "
  (cd /tmp/v16 && env $e timeout 600 python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(d['value'], 'images/s;', r.get('ms_per_forward_total'), 'ms per forward,', r.get('launches_total'), 'launches')")
done
(cd /tmp/v16 && BNDM_NCO64_MAX=256 timeout 900 python -m pytest tests/test_gpu_benched.py tests/test_gpu_unet.py -m gpu -q -x 2>&1 | tail -3)
echo "== v18 (conv_s16: 16-channel n-tiles for the conv1 launches of the 2x2 / 4x4 levels): accuracy + A/B on c2, then c5"
timeout 900 python tools/ab_libs.py --rounds 2 --acc bndm_amd/libbndm_hip.so tools/lib_v18.so "tools/lib_v18.so@BNDM_TAIL_N16_MAX=256" 2>&1 | tail -10
mkdir -p /tmp/v18 && cp -r bndm_amd tests oracle include bluenoise utils.py iadb_bn.py ddim_diffusers.py latent_iadb_bn_diffusers.py input_args.py pytest.ini __graft_entry__.py bench.py /tmp/v18/ 2>/dev/null && cp tools/lib_v18.so /tmp/v18/bndm_amd/libbndm_hip.so
(cd /tmp/v18 && timeout 600 python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300; timeout 900 python -m pytest tests/test_gpu_tail.py tests/test_gpu_benched.py -m gpu -q -x 2>&1 | tail -3)
echo "== v17 (dedicated head kernel + Euler epilogue, replaced round 5's v15): the loop / head tests of tests/ on the candidate, then a timed A/B"
mkdir -p /tmp/v17 && cp -r bndm_amd tests oracle include bluenoise utils.py iadb_bn.py ddim_diffusers.py latent_iadb_bn_diffusers.py input_args.py pytest.ini __graft_entry__.py bench.py /tmp/v17/ 2>/dev/null && cp tools/lib_v17.so /tmp/v17/bndm_amd/libbndm_hip.so
(cd /tmp/v17 && timeout 1200 python -m pytest tests/test_gpu_steps.py tests/test_gpu_tail.py tests/test_gpu_benched.py tests/test_gpu_unet.py tests/test_gpu_cli.py -m gpu -q -x 2>&1 | tail -4)
timeout 900 python tools/ab_libs.py --rounds 2 --full bndm_amd/libbndm_hip.so tools/lib_v17.so 2>&1 | tail -8
