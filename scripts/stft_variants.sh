#!/bin/bash
# build libshz.so with -DSTFT_OCC=<occ> and time the bench step at <wgs> stft workgroups per CU: "occ:wgs" pairs
for cfg in "$@"; do
  occ=${cfg%%:*}; wgs=${cfg##*:}
  touch shazam_amd/csrc/shz_extract.hip
  make -C shazam_amd/csrc EXTRA="-DSTFT_OCC=$occ $STFT_EXTRA" > /dev/null 2>&1 || { echo build failed; exit 1; }
  SHZ_STFT_WGS_PER_CU=$wgs timeout -k 10 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/var_bench.json 2> gpurun_out/var_bench.err
  python -c "
import json; d=json.load(open('gpurun_out/var_bench.json')); print('occ $occ wgs $wgs $STFT_EXTRA', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['kernel_ms_per_step'].items()})"
done
