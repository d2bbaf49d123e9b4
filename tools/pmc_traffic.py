#!/usr/bin/env python
"""HBM-side traffic per launch of the bench's kernel families -> profiles/pmc_traffic.json (read by bench.py's
`roofline.traffic`).

Input: the two rocprofv3 counter-collection CSVs of the SAME bench command, one pass per counter as
/opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/f -o b -- python bench.py --graph 0 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/w -o b -- python bench.py --graph 0 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events
    python tools/pmc_traffic.py out/f/b_counter_collection.csv out/w/b_counter_collection.csv profiles/pmc_traffic.json [libbts_amd.so md5]

The md5 of the profiled library goes into the file's `_meta` entry: bench.py only reports these figures as `roofline.traffic`
when it is timing the same binary (otherwise under `traffic_archived`).

Units and corrections: both counters are reported in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide
coalesced reads at 64 B, so it is doubled (same guide, HBM section); WRITE_SIZE is taken as is (uncalibrated there).
Infinity-Cache hits are counted by both, so this is fabric-side traffic, an upper bound of what reaches HBM.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def family(name):
    """Kernel symbol -> the family label bench.py uses (bts_amd/conv.py::_fwd_kernel / _wgrad_kernel, profiler labels)."""
    n = re.sub(r"\(anonymous namespace\)::|bts_conv::|^void ", "", name)
    n = re.sub(r"\(.*$", "", n)
    m = re.match(r"conv_igemm_dma<BF16, (\d), (\d), (\d), (\d)", n)
    if m:
        wr, wc, tm, tn = map(int, m.groups())
        return "conv_igemm_dma<bf16,%dx%d>" % (wr * tm * 32, wc * tn * 32)
    if n.startswith("conv_igemm_res<"):
        return "conv_igemm_res<bf16,128xN>"
    m = re.match(r"conv_wgrad_ring(?:_group)?<(\d), (\d)", n)      # the grouped launches are the same kernel body: one family
    if m:
        return "conv_wgrad_ring<bf16,%dx%d>" % (int(m.group(1)) * 64, int(m.group(2)) * 64)
    if n.startswith("conv_wgrad_tr"):
        return "conv_wgrad_tr<bf16,128x128>"
    m = re.match(r"conv_halo_wide<(\d)[,>]", n)
    if m:
        return "conv_halo_wide<bf16,%dx256>" % (int(m.group(1)) * 32)
    if n.startswith("conv_halo_wide"):
        return "conv_halo_wide<bf16,128x256>"
    if n.startswith("conv_wgrad_halo_tr_up"):
        return "conv_wgrad_halo_tr_up<bf16>"
    if n.startswith("conv_wgrad_halo_tr"):
        return "conv_wgrad_halo_tr<bf16>"
    if n.startswith("conv_c1_fwd"):
        return "conv_c1_fwd"
    if n.startswith("conv_c1_dgrad"):
        return "conv_c1_dgrad"
    if n.startswith("conv_c1_wgrad"):
        return "conv_c1_wgrad"
    if n.startswith("conv_wgrad_c1"):
        return "conv_wgrad_c1<bf16>"
    if n.startswith("bn_bwd_multi_"):
        return "bn_bwd_multi"
    if n.startswith("bn_bwd_"):
        return "bn_bwd"
    if n.startswith("bn_apply_ms"):
        return "bn_apply"
    if n.startswith("bn_stats_final_wide"):
        return "bn_stats_finalize"
    if n.startswith("bn_stats_"):
        return "bn_stats"
    if n.startswith("conv_halo<BF16"):
        return "conv_halo<bf16>"
    if n.startswith("conv_wgrad_halo_up"):
        return "conv_wgrad_halo_up<bf16>"
    if n.startswith("conv_wgrad_halo"):
        return "conv_wgrad_halo<bf16>"
    if n.startswith("lpg_chain_") or n.startswith("lpg_head_"):
        return "lpg_head*"
    return None


# families whose C-ABI call is several kernels: the CALL count is the dispatch count of one marker kernel (the counter bytes of all of
# the family's kernels are summed and divided by it, so the figure compares with bench.py's per-call algorithmic bytes)
CALL_MARKER = {"bn_bwd": "bn_bwd_apply", "bn_bwd_multi": "bn_bwd_multi_apply", "bn_stats": "bn_stats_partial"}


def load(path, counter):
    """family -> [sum of the counter over every kernel of the family, number of C-ABI calls]"""
    per = defaultdict(lambda: [0.0, 0])
    import gzip
    for r in csv.DictReader(gzip.open(path, "rt") if path.endswith(".gz") else open(path)):
        if r.get("Counter_Name") != counter:
            continue
        f = family(r["Kernel_Name"])
        if f:
            per[f][0] += float(r["Counter_Value"])
            n = re.sub(r"\(anonymous namespace\)::|bts_conv::|^void ", "", r["Kernel_Name"])
            if f not in CALL_MARKER or n.startswith(CALL_MARKER[f]):
                per[f][1] += 1
    return per


def alg_bytes(path):
    """family -> launch-weighted mean algorithmic bytes per C-ABI launch, from a `bench.py --dump-launches` table taken in the SAME
    configuration as the counter passes (BTS_CONV_WIDE=0)."""
    import json as _json
    acc = defaultdict(lambda: [0.0, 0.0])
    with open(path) as f:
        for r in _json.load(f)["rows"]:
            nb = r.get("alg_bytes_per_launch") or 0.0
            if nb > 0:
                fam = "lpg_head*" if r["family"].startswith("lpg_head") or r["family"].startswith("lpg_chain") else r["family"]
                acc[fam][0] += nb * r["launches_per_step"]
                acc[fam][1] += r["launches_per_step"]
    return {k: v[0] / v[1] for k, v in acc.items() if v[1] > 0}


def main():
    fpath, wpath, out = sys.argv[1:4]
    md5 = sys.argv[4] if len(sys.argv) > 4 else None
    alg = alg_bytes(sys.argv[5]) if len(sys.argv) > 5 else {}
    fetch, write = load(fpath, "FETCH_SIZE"), load(wpath, "WRITE_SIZE")
    table = {}
    for fam in sorted(set(fetch) & set(write)):
        nf, nw = fetch[fam][1], write[fam][1]
        if nf == 0 or nw == 0:
            continue
        rd = 2.0 * 1024.0 * fetch[fam][0] / nf
        wr = 1024.0 * write[fam][0] / nw
        table[fam] = {"bytes_per_launch": round(rd + wr), "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                      "launches_fetch_pass": nf, "launches_write_pass": nw,
                      "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --graph 0; FETCH_SIZE x2 "
                                "(gfx950 tallies 128-B requests at 64 B), KiB -> bytes; fabric-side, Infinity-Cache hits included"}
        if fam in alg:
            table[fam]["alg_bytes_per_launch"] = round(alg[fam])
            table[fam]["traffic_over_algorithmic"] = round((rd + wr) / alg[fam], 2)
    mix = os.environ.get("PMC_ENV", "").strip()
    table["_meta"] = {"library_md5": md5,
                      "config": ("both passes with %s: kernels that rocprofv3 aborted on in rounds 3-5 are switched off, so conv_igemm_dma covers more "
                                 "launches per step than in the timed step" % mix) if mix else
                                "default switches: both passes on the kernel mix of the timed step (conv_halo_wide and conv_igemm_res dispatched; "
                                "rounds 3-5 needed BTS_CONV_WIDE=0 BTS_RES=0 here)",
                      "note": "every figure is per C-ABI CALL (bench.py's unit): bn_bwd = reduction + final + apply kernels of one call, "
                      "bn_stats = partial + final -- their kernels' counters are summed per call"}
    with open(out, "w") as f:
        json.dump(table, f, indent=1)
    for k, v in table.items():
        if k.startswith("_"):
            continue
        print("%-32s %8.1f MB/launch (read %.1f, write %.1f) over %d launches" % (k, v["bytes_per_launch"] / 1e6, v["read_bytes_per_launch"] / 1e6,
                                                                               v["write_bytes_per_launch"] / 1e6, v["launches_fetch_pass"]))


if __name__ == "__main__":
    main()
