#!/bin/bash
# rocprofv3 SQ counter passes over one profile-only bench run (counters only: no trace domains besides --kernel-trace)
# usage: tools/pmc_sq.sh TAG   -> gpurun_out/pmc_TAG/{a,b,c}, summary in gpurun_out/pmc_TAG.txt
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cmd="python $R/bench.py --profile-only --no-cpu-baseline"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU -d $out/a -- $cmd > $out/a.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_LDS_UNALIGNED_STALL -d $out/b -- $cmd > $out/b.log 2>&1
# effective clock under DVFS: GRBM_GUI_ACTIVE / kernel duration (the chip clocks to its power budget: MI355X_MICROARCH.md, "DVFS give-back")
rocprofv3 --output-format csv --kernel-trace --pmc GRBM_GUI_ACTIVE -d $out/c -- $cmd > $out/c.log 2>&1
python $R/tools/pmc_sum.py $out/a $out/b > $R/gpurun_out/pmc_$tag.txt
python $R/tools/pmc_clock.py $out/c >> $R/gpurun_out/pmc_$tag.txt 2>&1
find $out -type f -size +8M -delete; find $out -name "*.db" -delete
cat $R/gpurun_out/pmc_$tag.txt | head -80
