#!/bin/bash
# same-box A/B of builds of libshz.so on the headline step: bash scripts/ab_headline.sh lib1.so lib2.so ...  (each three times)
for rep in 1 2 3; do for l in "$@"; do
  SHZ_LIB=$(pwd)/$l python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 5 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms_per_step']
print('$l', round(d['value']), round(d['ms_per_step'],3), {a:round(b,3) for a,b in k.items()})"
done; done
