#!/bin/bash
# Promotion of a candidate that WON its A/B on a GPU (tools/gpu_candidates.sh) -- run on the build box:
#     bash tools/promote.sh conv_t32_shortcut_stages lib_v9
#  1. applies tools/experiments/<patch>.patch to the product sources and rebuilds bndm_amd/libbndm_hip.so;
#  2. the new library must equal the candidate library that was measured (device code objects, tools/device_code_hash.py);
#  3. prints the next step: the FULL GPU suite on the new library (tools/gpu_suite.sh), and only after it is green
#     `python tests/golden/make_launch_traces.py` (new device-code hashes + trace digests), then deletes patch and candidate.
# Nothing is committed by this script.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
P=tools/experiments/$1.patch; L=tools/$2.so
[ -f $P ] && [ -f $L ] || { echo "usage: tools/promote.sh <patch name> <candidate library name>"; exit 1; }
git diff --quiet -- bndm_amd include || { echo "product sources have uncommitted changes"; exit 1; }
git apply $P
make -C bndm_amd/csrc -j8 > /dev/null
a=$(python tools/device_code_hash.py bndm_amd/libbndm_hip.so | sed 's/.*all=//'); b=$(python tools/device_code_hash.py $L | sed 's/.*all=//')
if [ "$a" != "$b" ]; then echo "rebuilt product ($a) is not the measured candidate ($b): other patches must be re-based first"; git apply -R $P; make -C bndm_amd/csrc -j8 > /dev/null; exit 1; fi
echo "applied $P; product library == $L ($a)"
echo "next: gpurun --timeout 1800 -- 'bash tools/gpu_suite.sh r05'   (full suite on the new library)"
echo "then: python tests/golden/make_launch_traces.py && python -m pytest tests -q -m 'not gpu' && git rm $P && rm $L && bash tools/build_candidates.sh (re-base the rest)"
