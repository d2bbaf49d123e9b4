#!/usr/bin/env python
"""Merge two rocprofv3 counter-collection CSVs (one --pmc FETCH_SIZE pass, one --pmc WRITE_SIZE pass of the SAME
command) into a markdown table: consecutive launches of the same kernel and grid are one row, counters averaged
per launch.  FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 counts 128-byte requests as 64 B,
/opt/skills/guides/MI355X_MICROARCH.md, HBM section).

usage: pmc_table.py <fetch_counter_collection.csv> <write_counter_collection.csv>
"""
import csv
import re
import sys


def rows(path, counter):
    out = []
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*$", "", name)
        out.append((int(r["Dispatch_Id"]), name, int(r["Grid_Size"]), int(r["Workgroup_Size"]), float(r["Counter_Value"])))
    out.sort()
    return out


OURS = re.compile(r"^(conv_|lpg|bn_|affine|act_bwd|add_to|nchw|nhwc|silog|adamw)")


def main():
    f = [r for r in rows(sys.argv[1], "FETCH_SIZE") if OURS.match(r[1])]
    w = [r for r in rows(sys.argv[2], "WRITE_SIZE") if OURS.match(r[1])]
    assert len(f) == len(w) and all(x[1:4] == y[1:4] for x, y in zip(f, w)), "the two passes must run the same command"
    # the k-th launch is the same launch in both passes; launches of one probe case = same kernel, same grid and a
    # FETCH_SIZE within 2 % (zero-fill / unpack launches of the weight-gradient cases sit in between: look back 3 groups)
    g = []
    for (_, name, grid, wg, fv), (_, _, _, _, wv) in zip(f, w):
        hit = None
        for x in reversed(g[-3:]):
            if x[0] == name and x[1] == grid and abs(x[3][0] - fv) <= 0.02 * max(x[3][0], 1.0):
                hit = x
                break
        if hit is not None:
            hit[3].append(fv)
            hit[4].append(wv)
        else:
            g.append([name, grid, wg, [fv], [wv]])
    print("| kernel | grid threads | wg | launches | FETCH_SIZE KiB / launch | read MB (2x) | WRITE_SIZE KiB / launch | write MB |")
    print("|---|---|---|---|---|---|---|---|")
    for name, grid, wg, fv, wv in g:
        fk, wk = sum(fv) / len(fv), sum(wv) / len(wv)
        print("| %s | %d | %d | %d | %.0f | %.1f | %.0f | %.1f |" % (name, grid, wg, len(fv), fk, 2 * fk * 1024 / 1e6, wk, wk * 1024 / 1e6))


if __name__ == "__main__":
    main()
