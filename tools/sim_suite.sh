#!/bin/bash
# Every configuration of tests/gfx950sim/suite.py on the instruction-level simulator (the shipped library's machine code,
# executed on the CPU; no GPU involved) -> profiles/<tag>_sim_suite.log + profiles/<tag>_sim_kernel_coverage.txt (which kernels of the
# library ran, in which configurations).  About an hour and a half on 8 cores.
#   usage:  bash tools/sim_suite.sh r06 [name ...]
tag=${1:-rXX}; shift
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
names=${@:-all}
python -m tests.gfx950sim.suite --procs ${PROCS:-8} --work ${WORK:-/tmp/gfx950sim_work} --coverage profiles/${tag}_sim_kernel_coverage.txt $names 2>&1 | tee profiles/${tag}_sim_suite.log
