#!/bin/bash
# same-box A/B of builds of libshz.so (same ABI) on the 1M x 30 s match: bash scripts/ab_lib.sh lib1.so lib2.so ...  (each twice)
O=gpurun_out; mkdir -p $O
for rep in 1 2; do for l in "$@"; do
  SHZ_LIB=$(pwd)/$l timeout -k 10 400 python bench_db.py --songs 1000000 --queries 2000 --query-seconds 10 --snr 10 --match-batch 200 --finalize-every 100000 > $O/ab_lib.json || exit 1
  python - $O/ab_lib.json $l <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], {k:d[k] for k in ('value','p99_ms','qps','top1_accuracy')}, flush=True)
PY
done; done
