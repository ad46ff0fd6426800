#!/bin/bash
# same-box A/B of peak_pick32 variants on the headline step: bash scripts/ab_peak.sh "AHEAD OCC" ...   e.g. "1 4" "2 3" "2 4"
for rep in 1 2; do for v in "$@"; do
  set -- $v
  SHZ_PEAK_AHEAD=$1 SHZ_PEAK_OCC=$2 timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ahead/occ', '$v', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms_per_step']['peak_pick'],4))"
  set --
done; done
