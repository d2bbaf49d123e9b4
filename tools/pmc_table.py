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


def groups(rs):
    g = []
    for _, name, grid, wg, val in rs:
        if g and g[-1][0] == name and g[-1][1] == grid:
            g[-1][3].append(val)
        else:
            g.append([name, grid, wg, [val]])
    return g


def main():
    f, w = groups(rows(sys.argv[1], "FETCH_SIZE")), groups(rows(sys.argv[2], "WRITE_SIZE"))
    ours = re.compile(r"^(conv_|lpg|bn_|affine|act_bwd|add_to|nchw|nhwc|pack_|unpack_|silog|adamw)")
    f = [x for x in f if ours.match(x[0])]
    w = [x for x in w if ours.match(x[0])]
    print("| kernel | grid threads | wg | launches | FETCH_SIZE KiB / launch | read MB (2x) | WRITE_SIZE KiB / launch | write MB |")
    print("|---|---|---|---|---|---|---|---|")
    for a, b in zip(f, w):
        assert a[0] == b[0] and a[1] == b[1], (a[:2], b[:2])
        fk, wk = sum(a[3]) / len(a[3]), sum(b[3]) / len(b[3])
        print("| %s | %d | %d | %d | %.0f | %.1f | %.0f | %.1f |" % (a[0], a[1], a[2], len(a[3]), fk, 2 * fk * 1024 / 1e6, wk, wk * 1024 / 1e6))


if __name__ == "__main__":
    main()
