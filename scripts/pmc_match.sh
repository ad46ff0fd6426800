#!/bin/bash
# SQ counters of the match kernels (bench_db at 100k x 30 s tracks): gpurun_out/<tag>_pmc_match.json
TAG=${1:-q}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench_db.py --songs 100000 --queries 1000 --snr 10"
rm -rf $R/gpurun_out/pmc_a $R/gpurun_out/pmc_b
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $R/gpurun_out/pmc_a -o p --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/pmc_b -o p --output-format csv -- $B > /dev/null 2>&1
python3 $R/scripts/pmc_counters.py $R/gpurun_out/${TAG}_pmc_match.json "bench_db 100k x 30 s, 1,000 queries" $R/gpurun_out/pmc_a $R/gpurun_out/pmc_b > /dev/null
python3 - $R/gpurun_out/${TAG}_pmc_match.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,c in d['kernels'].items():
    if k.startswith(('vt_','sort_scatter32','sort_hist32','m_expand')):
        print(k, {a:(round(b,3) if b<100 else int(b)) for a,b in c.items() if a.startswith('frac') or a.endswith('per_wave') or a in('SQ_WAVES','launches_seen')})
PY
rm -rf $R/gpurun_out/pmc_a $R/gpurun_out/pmc_b
