set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r01g; mkdir -p $O
B="timeout -k 10 600 python bench_db.py"
$B --songs 20000 --queries 4000 --snr 10 --shards 1 > $O/bench_db_20k_shards1.json
$B --songs 20000 --queries 4000 --snr 10 --shards 8 > $O/bench_db_20k_shards8.json
$B --songs 100000 --queries 4000 --snr 10 > $O/bench_db_100k_snr10.json
$B --songs 100000 --queries 4000 --snr 10 --shards 4 > $O/bench_db_100k_snr10_shards4.json
$B --songs 100000 --queries 2000 --snr 10 --match-batch 200 --mixed-ingest 1000 > $O/bench_db_mixed_100k.json
$B --songs 1000000 --queries 2000 --query-seconds 10 --snr 10 --match-batch 200 --finalize-every 100000 > $O/bench_db_config5_1M_x_30s.json
$B --songs 1000000 --queries 2000 --query-seconds 10 --snr 10 --match-batch 200 --finalize-every 100000 --mixed-ingest 1000 > $O/bench_db_config5_mixed_1M.json
for f in $O/*.json; do echo $f; python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print({k:d[k] for k in ('value','p99_ms','qps','top1_accuracy')}, {k:round(v,3) for k,v in d['build'].items() if k in('seconds_total','finalize_s','insert_s','songs_per_s')}, d.get('mixed'))
PY
done
