#!/bin/bash
# does a sub-batch whose staged spectrogram fits the 256 MB Infinity Cache cut the HBM traffic of
# stft_psd + peak_pick32 (5.2x the algorithmic bytes in round 2), and does the step get shorter?
# For each workspace limit: step time (plain run), FETCH_SIZE and WRITE_SIZE (own rocprofv3 passes, --kernel-trace only).
# Run on the GPU box from the repo root: scripts/traffic_experiment.sh r03
set -e
TAG=${1:-r03}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for WS in 0 800 400 200 100; do
  B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --ws-limit-mb $WS"
  python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --ws-limit-mb $WS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'ws_limit_mb': $WS, 'ms_per_step': d['ms_per_step'], 'kernel_ms_per_step': d['roofline']['kernel_ms_per_step']}))" > $R/gpurun_out/${TAG}_traffic_ws${WS}_time.json
  rm -rf $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_f -o p --output-format csv -- $B > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_w -o p --output-format csv -- $B > /dev/null 2>&1
  python3 $R/scripts/pmc_traffic.py $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w $R/gpurun_out/${TAG}_traffic_ws${WS}_pmc.json "bench.py --steps 3 --warmup 1 --no-extras --ws-limit-mb $WS, MI355X" 644000 --sum-per-step 4
  cat $R/gpurun_out/${TAG}_traffic_ws${WS}_time.json
done
rm -rf $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
