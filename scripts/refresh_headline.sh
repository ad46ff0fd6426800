# headline evidence of the current build: PMC traffic (two passes), kernel stats, full bench line -> gpurun_out/<tag>_*
set -e
TAG=${1:-r01g}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_f -o p --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_w -o p --output-format csv -- $B > /dev/null 2>&1
python3 $R/scripts/pmc_traffic.py $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w $R/gpurun_out/${TAG}_pmc_traffic.json "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras, MI355X, round 1 final code"
rm -rf $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
mkdir -p $R/profiles_tmp && cp $R/gpurun_out/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profk -o k -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
cp $R/gpurun_out/profk/k_kernel_stats.csv $R/gpurun_out/${TAG}_bench_kernel_stats.csv
rm -rf $R/gpurun_out/profk $R/profiles_tmp
cd $R && python3 bench.py > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/bench.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench_full.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["kernel_ms_per_step"], d.get("two_contexts_one_gpu",{}).get("audio_s_per_s"), d["cpu_baseline"]["value"])
PY
head -4 gpurun_out/${TAG}_bench_kernel_stats.csv
