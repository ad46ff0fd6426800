#!/usr/bin/env python3
"""Summarise rocprofv3 counter passes into one tracked JSON: per kernel, the average per launch of every counter.

    rocprofv3 --pmc C1 C2 ... --kernel-trace -d gpurun_out/pmc_a -o p --output-format csv -- python3 bench.py ...
    python3 scripts/pmc_counters.py profiles/r02_pmc.json "<note>" gpurun_out/pmc_a gpurun_out/pmc_b ...

SQ_* cycle counters (WAVE_CYCLES, WAIT_*, ACTIVE_INST_*) count quad-cycles summed over waves (MI355X_MICROARCH.md);
ratios between them are what the summary is for: busy_valu = ACTIVE_INST_VALU / WAVE_CYCLES etc. are written next to
the raw averages when both operands are present."""
import collections
import csv
import json
import re
import sys


def main():
    out, note, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    for d in dirs:
        for r in csv.DictReader(open(f"{d}/p_counter_collection.csv")):
            name = re.sub(r"^void ", "", r["Kernel_Name"])
            name = re.sub(r"\(.*$", "", name)
            tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
            n[name][r["Counter_Name"]] += 1
    kernels = {}
    for k in sorted(tot):
        c = {cn: tot[k][cn] / n[k][cn] for cn in sorted(tot[k])}
        c["launches_seen"] = max(n[k].values())
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for cn, label in (("SQ_ACTIVE_INST_VALU", "frac_valu"), ("SQ_ACTIVE_INST_SCA", "frac_salu"),
                              ("SQ_ACTIVE_INST_LDS", "frac_lds_issue"), ("SQ_ACTIVE_INST_VMEM", "frac_vmem_issue"),
                              ("SQ_WAIT_ANY", "frac_wait_any"), ("SQ_WAIT_INST_ANY", "frac_wait_inst_any"),
                              ("SQ_WAIT_INST_LDS", "frac_wait_inst_lds"), ("SQ_ACTIVE_INST_ANY", "frac_active_any")):
                if cn in c:
                    c[label] = c[cn] / wc
        if c.get("SQ_LDS_IDX_ACTIVE"):
            c["lds_bank_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        if c.get("SQ_WAVES"):
            for cn in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
                       "SQ_INSTS_FLAT", "SQ_INSTS_SMEM"):
                if cn in c:
                    c[cn + "_per_wave"] = c[cn] / c["SQ_WAVES"]
        kernels[k] = c
    json.dump({"source": "rocprofv3 --pmc (one pass per directory), " + note,
               "units": "counter averages per launch; SQ cycle counters are quad-cycles summed over waves; frac_* = counter / "
                        "SQ_WAVE_CYCLES", "kernels": kernels}, open(out, "w"), indent=1)
    for k, c in kernels.items():
        if "stft" in k or "peak_pick" in k:
            print(k, {a: (round(b, 4) if isinstance(b, float) and b < 100 else b) for a, b in c.items()
                      if a.startswith(("frac", "lds_bank")) or a.endswith("per_wave")})


if __name__ == "__main__":
    main()
