#!/bin/bash
# table build + noisy-query match at the BASELINE config sizes (one MI355X) -> gpurun_out/<tag>_bench_db_*.json
set -e
TAG=${1:-r02}
O=gpurun_out; mkdir -p $O
B="timeout -k 10 900 python bench_db.py"
# configs[2]/[3] as written: 100k x 3 min tracks, 10k x 5 s queries at SNR 0 dB (recognizer_test.py:39-40)
$B --songs 100000 --seconds 180 --queries 10000 --snr 0 --match-batch 500 --chunk 500 --finalize-every 10000 > $O/${TAG}_bench_db_config3_4_100k_x_180s_snr0.json
$B --songs 100000 --queries 4000 --snr 10 > $O/${TAG}_bench_db_100k_snr10.json
$B --songs 100000 --queries 4000 --snr 0 > $O/${TAG}_bench_db_100k_snr0.json
$B --songs 100000 --queries 4000 --snr 10 --shards 4 > $O/${TAG}_bench_db_100k_snr10_shards4.json
$B --songs 100000 --queries 2000 --snr 10 --match-batch 200 --mixed-ingest 1000 > $O/${TAG}_bench_db_mixed_100k.json
$B --songs 1000000 --queries 2000 --query-seconds 10 --snr 10 --match-batch 200 --finalize-every 50000 --mixed-ingest 1000 > $O/${TAG}_bench_db_config5_mixed_1M.json
for f in $O/${TAG}_bench_db_*.json; do echo $f; python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print({k:d[k] for k in ('value','p99_ms','qps','top1_accuracy')}, {k:round(v,3) for k,v in d['build'].items() if k in('seconds_total','fingerprint_s','finalize_s','insert_s','songs_per_s')}, d.get('mixed'))
PY
done
