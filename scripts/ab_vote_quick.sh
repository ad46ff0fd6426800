#!/bin/bash
# quick timing of the default match path at 1M x 30 s and 100k x 30 s tracks
O=gpurun_out; mkdir -p $O
timeout -k 10 400 python bench_db.py --songs 1000000 --queries 2000 --query-seconds 10 --snr 10 --match-batch 200 --finalize-every 100000 > $O/vtq_1M.json
timeout -k 10 300 python bench_db.py --songs 100000 --queries 4000 --snr 10 > $O/vtq_100k.json
for f in $O/vtq_1M.json $O/vtq_100k.json; do python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], {k:d[k] for k in ('value','p99_ms','qps','top1_accuracy')})
PY
done
