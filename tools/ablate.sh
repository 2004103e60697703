#!/bin/bash
# profiling aid: per-op timings of the 64x64 fused convs under each ablation of the fused conv kernel
# usage: tools/ablate.sh [ablation codes...]   (default: 0 15)
mkdir -p gpurun_out
for a in ${@:-0 15}; do
  BNDM_ABLATE=$a BNDM_T32_TRACE=gpurun_out/t32_trace_$a.txt BNDM_PROFILE_DUMP=gpurun_out/abl_$a.txt python bench.py --profile-only --no-cpu-baseline > /dev/null 2>&1
  echo "ABL=$a: d0.conv1 (K=1152): $(grep 'down_blocks.0.resnets.0.conv1' gpurun_out/abl_$a.txt | awk '{print $2}')  up5.conv1 (K=2304): $(grep 'up_blocks.5.resnets.0.conv1 ' gpurun_out/abl_$a.txt | awk '{print $2}')  up4.ups: $(grep 'up_blocks.4.upsamplers' gpurun_out/abl_$a.txt | awk '{print $2}') d1.conv1(32x32): $(grep 'down_blocks.1.resnets.0.conv1' gpurun_out/abl_$a.txt | awk '{print $2}')"
done
