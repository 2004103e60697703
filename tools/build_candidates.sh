#!/bin/bash
# Builds the candidate libraries of tools/next_gpu_session.sh from the patches under tools/experiments/ WITHOUT touching the
# product sources (works in a scratch copy of bndm_amd/csrc).  One patch per library -- singles before stacks:
#   tools/lib_v9.so   conv_t32_shortcut_stages.patch            staged 1x1 (conv_shortcut) chunks          bit-identical by construction
#   tools/lib_v12.so  conv_t32_scalar_chunks_on_product.patch   chunk descriptors from scalar kernel args  bit-identical by construction
#   tools/lib_v8.so   round4_pairstats_sumsfirst_th32.patch     pair-granular sums, sums-first prologue, conv_t32<TH=32> behind BNDM_TH32_MIN
#   tools/lib_v16.so  conv_t32_nco64_small_batch.patch          64-channel n-tiles for small-batch handles (BNDM_NCO64_MAX, off by default)
#   tools/lib_v17.so  head_conv_kernel.patch                    dedicated head kernel + Euler epilogue (bit-equal loop; replaced round 5's head_euler_step.patch)
#   tools/lib_v18.so  conv_s16_small_grids.patch                16-channel conv_s n-tiles for under-filled grids (conv1 of the 2x2 / 4x4 ResnetBlocks)
#   tools/lib_v19.so  conv_s_padding_reads_banked.patch         conv_s: bank-conflict-free padding reads and cross-wave slabs   bit-identical by construction
#   tools/lib_lanes.so lanes.patch                              bndm_unet_set_lanes (host side only: same kernels)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d); mkdir -p $T/bndm_amd $T/include
cp -r $R/bndm_amd/csrc $T/bndm_amd/; cp $R/include/*.h $T/include/; rm -f $T/bndm_amd/csrc/*.o
cd $T && git init -q .
one() {   # one <patch> <library>
  git apply --include='bndm_amd/csrc/*' --include='include/*' $R/tools/experiments/$1
  make -C bndm_amd/csrc -j8 > /dev/null && cp bndm_amd/libbndm_hip.so $R/tools/$2
  git apply -R --include='bndm_amd/csrc/*' --include='include/*' $R/tools/experiments/$1
}
one conv_t32_shortcut_stages.patch lib_v9.so
one conv_t32_scalar_chunks_on_product.patch lib_v12.so
one round4_pairstats_sumsfirst_th32.patch lib_v8.so
one conv_t32_nco64_small_batch.patch lib_v16.so
one head_conv_kernel.patch lib_v17.so
one conv_s16_small_grids.patch lib_v18.so
one conv_s_padding_reads_banked.patch lib_v19.so
one lanes.patch lib_lanes.so
rm -rf $T; sha256sum $R/bndm_amd/libbndm_hip.so $R/tools/lib_*.so
