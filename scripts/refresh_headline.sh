#!/bin/bash
# headline evidence of the current build: counter passes (SQ, traffic), kernel stats, full bench line -> gpurun_out/<tag>_*
# (run on the GPU box from the repo root; copy what is to be judged into profiles/)
set -e
TAG=${1:-r02}
R=$(pwd)
bash scripts/pmc_passes.sh $TAG > gpurun_out/${TAG}_pmc.log 2>&1
cd $R && python3 bench.py > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/${TAG}_bench.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench_full.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["staged_frac"], d["roofline"]["kernel_ms_per_step"], d["cpu_baseline"]["value"], d["single_query"]["total_p50_ms"], d["match_1M"]["p50_ms"], d["match_1M"]["batch200"]["ms_per_query_p50"])
PY
head -6 gpurun_out/${TAG}_bench_kernel_stats.csv
