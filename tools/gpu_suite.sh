#!/bin/bash
# First GPU call of a round: the full `-m gpu` suite and smoke() on the library as shipped, with its sha256 in the log.
#   usage (on the GPU box, from the repo root):  bash tools/gpu_suite.sh r05  -> gpurun_out/r05_gpu_tests.log
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
log=gpurun_out/${tag}_gpu_tests.log
{ echo "library: $(sha256sum bndm_amd/libbndm_hip.so)"; echo "date: $(date -u +%FT%TZ)"; rocminfo 2>/dev/null | grep -m1 gfx; } > $log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 >> $log
echo "-- smoke" >> $log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> $log
cat $log
