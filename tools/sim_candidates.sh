#!/bin/bash
# Gate in front of any GPU minute: a candidate library (tools/lib_*.so, one patch of tools/experiments/ each; tools/build_candidates.sh)
# must reproduce the oracle on the instruction-level simulator (tests/gfx950sim) with zero hazards -> profiles/<tag>_sim_candidates.log.
# Equal `out` hashes = bit-identical results: the candidates that claim bit-identity by construction (v9, v12, v19) must print the
# product's hash for the same configuration; v17's fused loop must print the hash of its own step-by-step form.
#   usage:  bash tools/sim_candidates.sh r06 [name ...]        (default: every candidate; names: v9 v12 v19 v8 v16 v17 v18 lanes)
tag=${1:-rXX}; shift
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
names=${@:-v9 v12 v19 v8 v16 v17 v18 lanes}
log=profiles/${tag}_sim_candidates.log
W=${WORK:-/tmp/gfx950sim_work_cand}
run() { python -m tests.gfx950sim.suite --procs ${PROCS:-8} --work $W "$@" 2>&1 | grep -v "^library" ; }
{
echo "# $(date -u +%FT%TZ)  product: $(sha256sum bndm_amd/libbndm_hip.so | cut -c1-64)"
echo "##### product (reference hashes)"
run c2_t32x4 c2_iadb_loop
for n in $names; do
  lib=tools/lib_$n.so
  echo "##### $lib  $(sha256sum $lib | cut -c1-64)"
  case $n in
    v9|v12) run --lib $lib c2_t32x4 ;;                                      # must print the product's c2_t32x4 hash
    v19) run --lib $lib c2_t32x4 c5 bottom1x1 c2_bf16_t32x4 lat256 ;;           # must print the product's hashes (profiles/<tag>_sim_suite.log)
    v8)  run --lib $lib c2_t32x4 w64
         echo "## conv_t32<TH=32> (512-pixel tiles, 512 registers, 2 spill slots) on the 64x64 layers: BNDM_TH32_MIN=1"
         BNDM_TH32_MIN=1 run --lib $lib c2 ;;
    v16) echo "## c5 handle (max_batch 8), BNDM_NCO64_MAX=256: every non-head conv_t32 launch on <TH=8, N=64>"
         BNDM_NCO64_MAX=256 run --lib $lib c5
         echo "## c2 handle (max_batch 64), BNDM_NCO64_MAX=1000000: <TH=16, N=64>"
         BNDM_NCO64_MAX=1000000 run --lib $lib c2 ;;
    v17) run --lib $lib c2 c2_iadb_loop w64 c2_bf16_t32x4 c5 c4 c4_b1_handle lat256
         echo "## the same loop with BNDM_NO_STEP_FUSION=1 (separate iadb_step launch): must print the fused loop's hash"
         BNDM_NO_STEP_FUSION=1 run --lib $lib c2_iadb_loop ;;
    v18) run --lib $lib c2 c5 c2_bf16_t32x4 bottom1x1 ;;                     # conv_s16 on the conv1 launches of the 2x2 / 4x4 levels
    lanes) echo "## the in-engine IADB loop as 2 chains of launches (batch 2, one sample per chain); product at batch 2 for the hash:"
         SIM_BATCH=2 run c2_iadb_loop
         EXEC_LANES=2 SIM_BATCH=2 run --lib $lib c2_iadb_loop ;;
  esac
done
} 2>&1 | tee $log
