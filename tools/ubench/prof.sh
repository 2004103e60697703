#!/bin/bash
# rocprofv3 kernel-trace summary of one of the stand-alone microbenchmarks: tools/ubench/prof.sh noise_bench.bin [args]
R=${GRAFT_REPO_ROOT:-/root/repo}
bin=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ubprof
rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/ubprof -- $R/tools/ubench/$bin "$@" > /tmp/ubprof.log 2>&1
f=$(find /tmp/ubprof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %6s  avg %9.2f us  min %9.2f  max %9.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
