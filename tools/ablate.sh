#!/bin/bash
# profiling aid: per-op timings of the 64x64 fused convs under each ablation of the fused conv kernel.  The ablation
# kernels (parts of the work removed, WRONG results) are compiled only here, into tools/libbndm_ablate.so, with
# -DBNDM_ABLATION; the product library bndm_amd/libbndm_hip.so does not contain them.
# usage: tools/ablate.sh [ablation codes...]   (default: 0 15)     build step needs hipcc (run it before gpurun)
R=$(cd "$(dirname "$0")/.." && pwd)
if [ ! -f $R/tools/libbndm_ablate.so ] || [ "$1" = "--build" ]; then
  mkdir -p /tmp/bndm_ablate && cd $R/bndm_amd/csrc &&
  for f in core bluenoise steps unet_kernels unet_gn unet_conv32 unet_tail unet_f32 unet_engine; do
    x=""; [ $f = unet_conv32 ] && x="-fno-slp-vectorize"      # (as the Makefile builds it)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $x -DBNDM_ABLATION -I../../include -c $f.hip -o /tmp/bndm_ablate/$f.o || exit 1
  done && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/libbndm_ablate.so /tmp/bndm_ablate/*.o
  [ "$1" = "--build" ] && exit 0
fi
cd $R; mkdir -p gpurun_out
for a in ${@:-0 15}; do  # (BNDM_ABLATE_NW8=1 in the environment forces the 8-wave variant)
  BNDM_ABLATE=$a BNDM_T32_TRACE=gpurun_out/t32_trace_$a.txt BNDM_PROFILE_DUMP=gpurun_out/abl_$a.txt python tools/ablate_run.py > /dev/null 2>&1
  echo "ABL=$a: d0.conv1 (K=1152): $(grep 'down_blocks.0.resnets.0.conv1' gpurun_out/abl_$a.txt | awk '{print $2}')  up5.conv1 (K=2304): $(grep 'up_blocks.5.resnets.0.conv1 ' gpurun_out/abl_$a.txt | awk '{print $2}')  up4.ups: $(grep 'up_blocks.4.upsamplers' gpurun_out/abl_$a.txt | awk '{print $2}') d1.conv1(32x32): $(grep 'down_blocks.1.resnets.0.conv1' gpurun_out/abl_$a.txt | awk '{print $2}')"
done
