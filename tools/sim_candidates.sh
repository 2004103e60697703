#!/bin/bash
# Gate in front of any GPU minute: a candidate library (tools/lib_*.so, one patch of tools/experiments/ each) must reproduce the
# oracle on the instruction-level simulator with zero hazards -> profiles/<tag>_sim_candidates.log
#   usage:  bash tools/sim_candidates.sh r06 [lib ...]        (default: every tools/lib_*.so)
tag=${1:-rXX}; shift
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
libs=${@:-$(ls tools/lib_*.so)}
log=profiles/${tag}_sim_candidates.log
: > $log
for lib in $libs; do
  case $(basename $lib) in
    lib_v15.so) cfgs="c2_iadb_loop" ;;                      # the head's Euler epilogue only runs inside the in-engine loop
    lib_v8.so)  cfgs="c2_t32x4 c4" ;;                       # pair-granular sums everywhere (TH=32 itself needs >= 448 workgroups: see the log)
    *)          cfgs="c2_t32x4" ;;
  esac
  echo "##### $lib" | tee -a $log
  python -m tests.gfx950sim.suite --lib $lib --procs ${PROCS:-8} --work ${WORK:-/tmp/gfx950sim_work} $cfgs 2>&1 | tee -a $log
done
