#!/bin/bash
# Counter passes over the bench step (run on the GPU box from the repo root): scripts/pmc_passes.sh <tag>
# Each pass is its own rocprofv3 run (--pmc with --kernel-trace only).  Summaries land in gpurun_out/<tag>_*.json.
set -e
TAG=${1:-r02}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rm -rf $R/gpurun_out/pmc_*
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $R/gpurun_out/pmc_a -o p --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/pmc_b -o p --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM --kernel-trace -d $R/gpurun_out/pmc_c -o p --output-format csv -- $B > /dev/null 2>&1
python3 $R/scripts/pmc_counters.py $R/gpurun_out/${TAG}_pmc_counters.json "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras, MI355X" $R/gpurun_out/pmc_a $R/gpurun_out/pmc_b $R/gpurun_out/pmc_c
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_f -o p --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_w -o p --output-format csv -- $B > /dev/null 2>&1
python3 $R/scripts/pmc_traffic.py $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w $R/gpurun_out/${TAG}_pmc_traffic.json "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras, MI355X" ${FRAMES_PER_LAUNCH:-644000}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
cp $R/gpurun_out/prof_stats/p_kernel_stats.csv $R/gpurun_out/${TAG}_bench_kernel_stats.csv
rm -rf $R/gpurun_out/pmc_? $R/gpurun_out/prof_stats
