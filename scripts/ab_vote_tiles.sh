#!/bin/bash
# A/B of the vote-tile fold against the full sort at 1M x 30 s and 100k x 30 s tracks, then a kernel trace of the match
# phase with tiles -> gpurun_out/vt_*.json, gpurun_out/vt_trace/
set -e
O=gpurun_out; mkdir -p $O
A="--songs 1000000 --queries 2000 --query-seconds 10 --snr 10 --match-batch 200 --finalize-every 100000"
SHZ_VOTE_TILES=0 timeout -k 10 400 python bench_db.py $A > $O/vt_1M_sort.json
timeout -k 10 400 python bench_db.py $A > $O/vt_1M_tiles.json
SHZ_VOTE_TILES=0 timeout -k 10 300 python bench_db.py --songs 100000 --queries 4000 --snr 10 > $O/vt_100k_sort.json
timeout -k 10 300 python bench_db.py --songs 100000 --queries 4000 --snr 10 > $O/vt_100k_tiles.json
for f in $O/vt_1M_sort.json $O/vt_1M_tiles.json $O/vt_100k_sort.json $O/vt_100k_tiles.json; do python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], {k:d[k] for k in ('value','p99_ms','qps','top1_accuracy')})
PY
done
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/$O/vt_trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/vt_trace -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench_db.py --songs 1000000 --queries 600 --query-seconds 10 --snr 10 --match-batch 200 --finalize-every 100000 > $GRAFT_REPO_ROOT/$O/vt_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/trace_after.py $O/vt_trace/t_kernel_trace.csv m_probe_kernel > $O/vt_match_1M_kernel_stats.csv
head -30 $O/vt_match_1M_kernel_stats.csv
