#!/bin/bash
# Differential run on the instruction-level simulator (a debugging aid): every launch of an exec_forward.py case is executed on the simulator,
# its stores are compared with the numpy contract model of the same launch, and memory continues with the MODEL's result -- so each launch is
# judged on clean inputs and the first kernel that disagrees is named, down to the byte offset.
#   usage: bash tools/sim_diff.sh <case> [batch]          e.g.  bash tools/sim_diff.sh w64 1
#   env:   SIMLIB=<lib.so>            another library (a candidate)
#          GFX950SIM_ONLY=2,20|name   simulate only these launch indices / kernels whose name contains the text (the rest run as models: fast)
#          GFX950SIM_WGSAMPLE=4       only 4 workgroups of each launch (launches at the benchmark's batch for the price of a few workgroups)
#          EXEC_MAX_BATCH=64          size the handle for another batch (tile variants follow it)
case=$1; batch=${2:-1}
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
W=${WORK:-/tmp/gfx950sim_work_diff}; mkdir -p $W/$case
lib=${SIMLIB:-bndm_amd/libbndm_hip.so}
read -r wfile kargs <<< $(python - "$case" "$W" "$lib" <<'PY'
import sys
sys.path.insert(0, ".")
from tests import test_launch_trace as T
from tests.hipmock import harness as H
case, work, lib = sys.argv[1:4]
H.build_mock(work)
cfg, key, make = T._case_network(case)
sd, wfile = T.oracle_weights(work, key, make)
print(wfile, H.kernargs_file(lib, work))
PY
)
LD_LIBRARY_PATH=$W:$LD_LIBRARY_PATH HIPMOCK_TRACE=$W/$case/trace.txt HIPMOCK_KERNARGS=$kargs EXEC_SIM=diff EXEC_SIM_VERBOSE=1 EXEC_BATCH=$batch \
  python tests/hipmock/exec_forward.py $lib $W/$case $case $wfile
