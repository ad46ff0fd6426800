#!/bin/bash
# match-side evidence of the current build -> gpurun_out/<tag>_*: bench_db at the BASELINE config sizes, per-kernel totals of
# the match phase at 1M x 30 s tracks, the dispatches of one 5 s query (run on the GPU box from the repo root)
set -e
TAG=${1:-r02}
R=$(pwd)
O=$R/gpurun_out
bash scripts/refresh_bench_db.sh $TAG > $O/${TAG}_bench_db.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/tr1 $O/tr2
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/tr1 -o t --output-format csv -- python3 $R/bench_db.py --songs 1000000 --queries 600 --query-seconds 10 --snr 10 --match-batch 200 --finalize-every 100000 > $O/${TAG}_trace_1M.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace -d $O/tr2 -o t --output-format csv -- python3 $R/scripts/single_query_trace.py 1000 20 > $O/${TAG}_trace_sq.log 2>&1
cd $R
python3 scripts/trace_after.py $O/tr1/t_kernel_trace.csv m_probe_kernel > $O/${TAG}_match_1M_kernel_stats.csv
python3 scripts/trace_gaps.py $O/tr2/t_kernel_trace.csv 45 > $O/${TAG}_single_query_trace.txt
python3 scripts/single_query_trace.py 1000 200 >> $O/${TAG}_single_query_trace.txt
rm -rf $O/tr1 $O/tr2
tail -14 $O/${TAG}_bench_db.log
head -14 $O/${TAG}_match_1M_kernel_stats.csv
tail -3 $O/${TAG}_single_query_trace.txt
