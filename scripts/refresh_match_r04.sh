#!/bin/bash
# match-side evidence of the current build -> gpurun_out/<tag>_* (run on the GPU box from the repo root; copy what is to be
# judged into profiles/): bench_db at the BASELINE config sizes, per-kernel totals of the match phase at 1M x 30 s tracks,
# SQ counters and FETCH_SIZE / WRITE_SIZE of the match kernels at 1M songs, the dispatches of one query.
TAG=${1:-r04}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
bash scripts/refresh_bench_db.sh $TAG > $O/${TAG}_bench_db.log 2>&1
echo "bench_db done"; tail -14 $O/${TAG}_bench_db.log
cd /tmp && export TMPDIR=/tmp
B1M="python3 $R/bench_db.py --songs 1000000 --queries 600 --query-seconds 10 --snr 10 --match-batch 200 --finalize-every 50000"
rm -rf $O/tr1 $O/tr2 $O/pmc_a $O/pmc_b $O/pmc_f $O/pmc_w
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/tr1 -o t --output-format csv -- $B1M > $O/${TAG}_trace_1M.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tr2 -o t --output-format csv -- python3 $R/scripts/single_query_1m_trace.py 1000000 20 > $O/${TAG}_trace_sq1m.log 2>&1
cd $R
python3 scripts/trace_after.py $O/tr1/t_kernel_trace.csv m_probe_kernel > $O/${TAG}_match_1M_kernel_stats.csv
python3 scripts/trace_gaps.py $O/tr2/t_kernel_trace.csv 34 > $O/${TAG}_single_query_1M_trace.txt
tail -1 $O/${TAG}_trace_sq1m.log >> $O/${TAG}_single_query_1M_trace.txt
rm -rf $O/tr1 $O/tr2
echo "traces done"; head -12 $O/${TAG}_match_1M_kernel_stats.csv
cd /tmp
timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $O/pmc_a -o p --output-format csv -- $B1M > /dev/null 2>&1
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace -d $O/pmc_b -o p --output-format csv -- $B1M > /dev/null 2>&1
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f -o p --output-format csv -- $B1M > /dev/null 2>&1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w -o p --output-format csv -- $B1M > /dev/null 2>&1
cd $R
python3 scripts/pmc_counters.py $O/${TAG}_pmc_match.json "bench_db 1M x 30 s, 600 x 10 s queries in batches of 200" $O/pmc_a $O/pmc_b > /dev/null
python3 scripts/pmc_traffic.py $O/pmc_f $O/pmc_w $O/${TAG}_pmc_match_traffic.json "bench_db 1M x 30 s, 600 x 10 s queries in batches of 200 (match kernels: per launch = one vote pass of ~2.4e8 votes)" 1
rm -rf $O/pmc_a $O/pmc_b $O/pmc_f $O/pmc_w
python3 - $O/${TAG}_pmc_match.json $O/${TAG}_pmc_match_traffic.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,c in d['kernels'].items():
    if k.startswith(('vt_','sort_scatter32','sort_hist32','m_expand','m_probe')):
        print(k, {a:(round(b,3) if b<100 else int(b)) for a,b in c.items() if a.startswith('frac') or a.endswith('per_wave') or a in('SQ_WAVES','launches_seen')})
t=json.load(open(sys.argv[2]))
for k,c in t.get('kernels',{}).items():
    if k.startswith(('vt_','sort_scatter32','sort_hist32','m_expand','m_probe')):
        print(k, c)
PY
