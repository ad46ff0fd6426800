#!/bin/bash
# two SQ counter passes over the bench step; summary to gpurun_out/<tag>_pmc_counters.json
TAG=${1:-q}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rm -rf $R/gpurun_out/pmc_a $R/gpurun_out/pmc_b
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $R/gpurun_out/pmc_a -o p --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/pmc_b -o p --output-format csv -- $B > /dev/null 2>&1
python3 $R/scripts/pmc_counters.py $R/gpurun_out/${TAG}_pmc_counters.json "quick" $R/gpurun_out/pmc_a $R/gpurun_out/pmc_b
rm -rf $R/gpurun_out/pmc_a $R/gpurun_out/pmc_b
