#!/bin/bash
# same-box A/B of SHZ_STFT_OPT values on the headline step: bash scripts/ab_stft.sh 0 1 2 3
for rep in 1 2; do for v in "$@"; do
  SHZ_STFT_OPT=$v timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('opt', $v, round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms_per_step']['stft_psd'],4))"
done; done
