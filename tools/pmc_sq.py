#!/usr/bin/env python
"""Summarise a rocprofv3 SQ counter pass (one --pmc run of bench.py) per kernel family: counter sums per launch, and the derived
figures the judge asks for -- MFMA-busy fraction of the wave cycles, issue-stall and wait fractions.

    python tools/pmc_sq.py <counter_collection.csv[.gz]> <out.json> [library_md5] [launches.json]

With a `bench.py --dump-launches` table of the SAME configuration (BTS_CONV_WIDE=0): per family the MFMA-pipe utilisation
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch duration x 2.4 GHz) -- the counter is in shader cycles, 32 per v_mfma_f32_32x32x16_bf16
(MI355X_MICROARCH.md); 2.4 GHz is the maximum clock, so the figure is a LOWER bound of the busy fraction at the clock the kernel really ran at --
and the executed / algorithmic FLOP ratio (tile padding: busy cycles / 32 x 32768 FLOPs against the launch's algorithmic FLOPs).

Families are the ones bench.py / tools/pmc_traffic.py use (pmc_traffic.family)."""
import csv
import json
import os
import sys
from collections import defaultdict

from pmc_traffic import family


def main():
    import gzip
    path, out = sys.argv[1:3]
    md5 = sys.argv[3] if len(sys.argv) > 3 else None
    launches = sys.argv[4] if len(sys.argv) > 4 else None
    per = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    fh_in = gzip.open(path, "rt") if path.endswith(".gz") else open(path)
    for r in csv.DictReader(fh_in):
        f = family(r["Kernel_Name"])
        if not f:
            continue
        per[f][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[f].add(r["Dispatch_Id"])
    table = {}
    for f, c in sorted(per.items()):
        n = max(1, len(disp[f]))
        row = {"launches": n}
        row.update({k: round(v / n, 1) for k, v in sorted(c.items())})
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc > 0:
            # SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD-quad-cycle units on gfx950's rocprofv3: report it against SQ_BUSY_CYCLES too
            for name, key in (("mfma_busy_per_wave_cycle", "SQ_VALU_MFMA_BUSY_CYCLES"), ("wait_any_frac", "SQ_WAIT_ANY"),
                              ("wait_inst_any_frac", "SQ_WAIT_INST_ANY"), ("active_inst_any_frac", "SQ_ACTIVE_INST_ANY")):
                if key in c:
                    row[name] = round(c[key] / wc, 4)
        if c.get("SQ_BUSY_CYCLES", 0.0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            row["mfma_busy_per_sq_busy_cycle"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"], 4)
        if c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16") and c.get("SQ_BUSY_CYCLES"):
            row["mfma_mops_bf16_per_launch"] = round(c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] / n, 1)
        table[f] = row
    if launches:
        fam = {}
        with open(launches) as fh:
            for r in json.load(fh)["rows"]:
                a = fam.setdefault(r["family"], [0.0, 0.0, 0.0])
                a[0] += r["launches_per_step"]
                a[1] += r["us_per_step"]
                a[2] += r["work_per_launch"] * r["launches_per_step"]
        for f, row in table.items():
            busy = row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            if f in fam and busy > 0 and fam[f][0] > 0:
                us, flops = fam[f][1] / fam[f][0], fam[f][2] / fam[f][0]
                row["event_timed_us_per_launch"] = round(us, 1)
                row["mfma_util_at_2.4GHz"] = round(busy / (1024 * us * 1e-6 * 2.4e9), 4)
                row["executed_over_algorithmic_flops"] = round(busy / 32 * 32768 / flops, 3)
    table["_meta"] = {"library_md5": md5, "source": "rocprofv3 --pmc (one SQ pass) --kernel-trace of bench.py --graph 0 (kernel mix: PMC_ENV of tools/final_protocol.sh = '%s'; "
                      "empty = the default switches of the timed step)" % os.environ.get("PMC_ENV", "")}
    with open(out, "w") as fh:
        json.dump(table, fh, indent=1)
    for f, row in table.items():
        if not f.startswith("_"):
            print("%-34s %s" % (f, {k: v for k, v in row.items() if k.endswith("frac") or k.startswith("mfma_") or k.startswith("executed") or k == "launches"}))


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main()
