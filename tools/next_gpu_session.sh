#!/bin/bash
# What to run when a GPU is available, in this order, ONE gpurun call per step (each is time-boxed on its own):
#
#   0. on the build box:  make -C bndm_amd/csrc && bash tools/build_candidates.sh && bash tools/ubench/build.sh
#      (no GPU needed, and a gate for step 3: bash tools/sim_suite.sh rNN; bash tools/sim_candidates.sh rNN -- a candidate whose machine
#      code does not reproduce the oracle on the instruction-level simulator with zero hazards gets no GPU minutes)
#   1. gpurun --timeout 1800 -- 'bash tools/gpu_suite.sh r06'
#        the full `-m gpu` suite + smoke() on the SHIPPED library, sha256 in the log -> copy gpurun_out/r06_gpu_tests.log to profiles/
#   2. gpurun --timeout 1800 -- 'bash tools/profile_round.sh r06'
#        PMC traffic keyed to the library's sha, bench line with per-op HIP events, rocprofv3 kernel stats, SQ counters
#   3. gpurun --timeout 2400 -- 'bash tools/gpu_candidates.sh'            (time-box: 60 GPU-minutes over all its runs)
#        single-patch candidate libraries, bit-identical ones first: hash against the shipped library, one interleaved A/B
#        round each.  Promotion rule: a winner's patch is applied to the product sources, the FULL suite (step 1) runs on the
#        new library, tests/golden/make_launch_traces.py is re-run, patch + candidate .so are deleted; a loser gets one line
#        in DESIGN's lever table and is deleted.
#   4. gpurun --timeout 2400 -- 'bash tools/gpu_lanes.sh'                 (time-box: 40 GPU-minutes)
#        chains of launches inside one GPU: first the form that needs NO library change (one engine handle + host thread per
#        chain, tools/two_stream.py), then -- in the box's scratch copy only -- tools/experiments/lanes.patch with its parked tests.
#   5. microbenchmarks that answer open DESIGN questions (tools/ubench: t32_bench grid sweep, anyorder, flagwait), last.
echo "see the header of this file: one gpurun call per step"; sed -n 2,20p "$0"
