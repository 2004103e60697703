#!/bin/bash
# Builds the candidate libraries of tools/next_gpu_session.sh from the patches under tools/experiments/ WITHOUT touching the
# product sources (works in a scratch copy of bndm_amd/csrc):  tools/lib_v6.so (pair-granular GroupNorm sums + sums-first
# prologue + TH=32 path, = the round-4 patch), tools/lib_v8.so (the same library, TH=32 selected by BNDM_TH32_MIN at run time),
# tools/lib_v7.so (+ scalar chunk descriptors).
# tools/lib_v9.so: the product sources + conv_t32's staged 1x1 (shortcut) chunks (conv_t32_shortcut_stages.patch) -- bit-identical
# to the shipped library by construction (tools/fwd_hash.py).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d); mkdir -p $T/bndm_amd $T/include
cp -r $R/bndm_amd/csrc $T/bndm_amd/; cp $R/include/*.h $T/include/; rm -f $T/bndm_amd/csrc/*.o
cd $T && git init -q . && git apply $R/tools/experiments/conv_t32_shortcut_stages.patch && make -C bndm_amd/csrc -j8 > /dev/null && cp bndm_amd/libbndm_hip.so $R/tools/lib_v9.so
git apply -R $R/tools/experiments/conv_t32_shortcut_stages.patch
# tools/lib_v12.so: the product sources + scalar chunk descriptors alone (bit-identical to the shipped library as well)
git apply $R/tools/experiments/conv_t32_scalar_chunks_on_product.patch && make -C bndm_amd/csrc -j8 > /dev/null && cp bndm_amd/libbndm_hip.so $R/tools/lib_v12.so
git apply -R $R/tools/experiments/conv_t32_scalar_chunks_on_product.patch
# tools/lib_v13.so: the product sources + write-back stores for conv_t32 workgroups that are not in the last round (bit-identical)
git apply $R/tools/experiments/conv_t32_first_round_write_back.patch && make -C bndm_amd/csrc -j8 > /dev/null && cp bndm_amd/libbndm_hip.so $R/tools/lib_v13.so
# tools/lib_v14.so: the three bit-identical candidates stacked on the product sources (v9 + v12 + v13)
git apply $R/tools/experiments/conv_t32_shortcut_stages.patch && git apply $R/tools/experiments/conv_t32_scalar_chunks_on_product.patch
make -C bndm_amd/csrc -j8 > /dev/null && cp bndm_amd/libbndm_hip.so $R/tools/lib_v14.so
git apply -R $R/tools/experiments/conv_t32_scalar_chunks_on_product.patch && git apply -R $R/tools/experiments/conv_t32_shortcut_stages.patch
git apply -R $R/tools/experiments/conv_t32_first_round_write_back.patch
git apply $R/tools/experiments/round4_pairstats_sumsfirst_th32.patch
make -C bndm_amd/csrc -j8 > /dev/null && cp bndm_amd/libbndm_hip.so $R/tools/lib_v8.so && cp bndm_amd/libbndm_hip.so $R/tools/lib_v6.so
git apply $R/tools/experiments/conv_t32_scalar_chunks.patch && make -C bndm_amd/csrc -j8 > /dev/null && cp bndm_amd/libbndm_hip.so $R/tools/lib_v7.so
# tools/lib_v11.so: everything stacked (round-4 patch + scalar chunk descriptors + staged 1x1 chunks)
git apply $R/tools/experiments/conv_t32_shortcut_stages.patch && make -C bndm_amd/csrc -j8 > /dev/null && cp bndm_amd/libbndm_hip.so $R/tools/lib_v11.so
rm -rf $T; ls -la $R/tools/lib_v[6789].so $R/tools/lib_v11.so $R/tools/lib_v12.so $R/tools/lib_v13.so $R/tools/lib_v14.so
