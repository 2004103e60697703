#!/bin/bash
# One-stop profile capture on the GPU box: tools/profile_round.sh r02  -> gpurun_out/<tag>_* (copy what is judged to profiles/)
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out profiles
python tools/pmc_traffic.py > gpurun_out/${tag}_pmc_traffic.log 2>&1
BNDM_PROFILE_DUMP=gpurun_out/${tag}_per_op_hipevents.txt python bench.py --steps 3 --warmup 1 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
# per-op HIP-event tables of the other configurations at their per-GPU sizes (c4: 128 px B=32, c5: latent B=8, c3: DDIM B=64)
for c in c3 c4 c5; do
  BNDM_PROFILE_DUMP=gpurun_out/${tag}_per_op_hipevents_${c}.txt timeout 600 python bench.py --config $c --profile-only --no-cpu-baseline > gpurun_out/${tag}_profile_${c}.json 2>> gpurun_out/${tag}_bench.err
done
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/${tag}_kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $R/gpurun_out/${tag}_kt.log 2>&1)
python tools/kstats.py gpurun_out/${tag}_kt > gpurun_out/${tag}_kernel_stats.txt
cp $(find gpurun_out/${tag}_kt -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/${tag}_kt
# the same for c5's per-GPU share (latent UNet at B = 8 + VAE decode): where a launch-bound forward spends its time
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/${tag}_kt5 -- python $R/bench.py --config c5 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_kt5.log 2>&1)
python tools/kstats.py gpurun_out/${tag}_kt5 > gpurun_out/${tag}_kernel_stats_c5.txt 2>/dev/null
rm -rf gpurun_out/${tag}_kt5
bash tools/pmc_sq.sh ${tag} > /dev/null 2>&1
rm -rf gpurun_out/pmc_${tag}
tail -c 2500 gpurun_out/${tag}_bench.json; echo; head -12 gpurun_out/${tag}_kernel_stats.txt
