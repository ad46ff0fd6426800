#!/usr/bin/env python3
"""Turn two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) into profiles/<tag>_pmc_traffic.json.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_f -o p --output-format csv -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_w -o p --output-format csv -- python3 bench.py ...
    python3 scripts/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/r01d_pmc_traffic.json "<note>"

Per kernel: the average per launch of each counter (KiB x 1024 = bytes) and
hbm_bytes_corrected = 2 x FETCH_SIZE + WRITE_SIZE -- MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies the 128-B
requests of wide coalesced reads at 64 B; narrow loads are uncalibrated, so the raw sum is kept next to it."""
import collections
import csv
import json
import re
import sys


def per_launch(dirname, counter, totals=None):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f"{dirname}/p_counter_collection.csv")):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = re.sub(r"\(.*$", "", name)                       # drop the argument list
        tot[name] += float(r["Counter_Value"]) * 1024.0
        n[name] += 1
    if totals is not None:
        totals.update({k: (tot[k], n[k]) for k in tot})
    return {k: tot[k] / n[k] for k in tot}


def main():
    fdir, wdir, out, note = sys.argv[1], sys.argv[2], sys.argv[3], (sys.argv[4] if len(sys.argv) > 4 else "")
    frames_per_launch = int(sys.argv[5]) if len(sys.argv) > 5 and not sys.argv[5].startswith("--") else 644000
    steps = int(sys.argv[sys.argv.index("--sum-per-step") + 1]) if "--sum-per-step" in sys.argv else 0   # passes over the batch in the run
    ft, wt = {}, {}
    f, w = per_launch(fdir, "FETCH_SIZE", ft), per_launch(wdir, "WRITE_SIZE", wt)
    kernels = {}
    for k in sorted(set(f) | set(w)):
        fr, wr = f.get(k, 0.0), w.get(k, 0.0)
        kernels[k] = {"fetch_raw": fr, "write": wr, "hbm_bytes_corrected": 2 * fr + wr, "hbm_bytes_raw": fr + wr}
        if steps:   # sub-batched runs launch a kernel several times per pass: bytes per pass over the whole batch
            kernels[k]["launches_per_step"] = ft.get(k, wt.get(k, (0, 0)))[1] / steps
            kernels[k]["hbm_bytes_corrected_per_step"] = (2 * ft.get(k, (0.0, 0))[0] + wt.get(k, (0.0, 0))[0]) / steps
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), " + note,
               "units": "bytes per launch (KiB counters x 1024); hbm_bytes_corrected = 2 x FETCH_SIZE + WRITE_SIZE per "
                        "MI355X_MICROARCH.md (gfx950 FETCH_SIZE tallies 128-B requests at 64 B); narrow loads are "
                        "uncalibrated, so raw and corrected are both kept",
               "frames_per_launch": frames_per_launch, "kernels": kernels}, open(out, "w"), indent=1)
    for k in ("stft_psd_kernel<float>", "peak_pick32_kernel<2, 4>"):
        if k in kernels:
            print(k, {a: round(b / 1e9, 3) for a, b in kernels[k].items()}, "GB")


if __name__ == "__main__":
    main()
