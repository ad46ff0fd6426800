#!/bin/bash
# same-box A/B of one environment variable on the 1M x 30 s match: bash scripts/ab_env.sh NAME v1 v2 ...   (each value twice)
O=gpurun_out; mkdir -p $O
N=$1; shift
for rep in 1 2; do for v in "$@"; do
  env $N=$v timeout -k 10 400 python bench_db.py --songs 1000000 --queries 2000 --query-seconds 10 --snr 10 --match-batch 200 --finalize-every 100000 > $O/ab_$v.json || exit 1
  python - $O/ab_$v.json $N $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], {k:d[k] for k in ('value','p99_ms','qps','top1_accuracy')}, flush=True)
PY
done; done
