#!/bin/bash
# what the GPU clocks and draws while the headline step runs: rocm-smi sampled beside `bench.py --steps N`
O=gpurun_out/clock; mkdir -p $O
python bench.py --steps ${1:-2500} --warmup 20 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err &
BP=$!
sleep 12
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|busy" | tr '\n' ' ' >> $O/smi.log; echo >> $O/smi.log
  sleep 1
done
wait $BP
tail -c 400 $O/bench.json; echo; cat $O/smi.log
