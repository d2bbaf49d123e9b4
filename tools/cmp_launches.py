"""Per-launch A/B of two (or 2 x n interleaved) `bench.py --dump-launches` tables: rows whose time differs by more than --min-us or
--min-pct, and the per-family totals.

    python tools/cmp_launches.py A1.json[,A2.json...] B1.json[,B2.json...] [--min-us 2] [--min-pct 3]
"""
import argparse
import json


def load(paths):
    acc = {}
    for p in paths.split(","):
        with open(p) as f:
            d = json.load(f)
        for r in d["rows"]:
            k = (r["family"], r["tag"])
            a = acc.setdefault(k, [0.0, 0, r["launches_per_step"]])
            a[0] += r["us_per_step"]
            a[1] += 1
    return {k: (v[0] / v[1], v[2]) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--min-us", type=float, default=2.0)
    ap.add_argument("--min-pct", type=float, default=3.0)
    args = ap.parse_args()
    A, B = load(args.a), load(args.b)
    tagsA = {}
    for (fam, tag), (us, n) in A.items():
        tagsA.setdefault(tag or fam, []).append((fam, us))
    tagsB = {}
    for (fam, tag), (us, n) in B.items():
        tagsB.setdefault(tag or fam, []).append((fam, us))
    rows = []
    for t in sorted(set(tagsA) | set(tagsB)):
        ua = sum(u for _, u in tagsA.get(t, []))
        ub = sum(u for _, u in tagsB.get(t, []))
        fa = "+".join(f for f, _ in tagsA.get(t, [])) or "-"
        fb = "+".join(f for f, _ in tagsB.get(t, [])) or "-"
        rows.append((ub - ua, t, fa, fb, ua, ub))
    rows.sort()
    print("%-52s %-34s %9s %9s %8s" % ("launch", "family (B if it differs)", "A us", "B us", "delta"))
    for d, t, fa, fb, ua, ub in rows:
        if abs(d) >= args.min_us and (ua == 0 or abs(d) / max(ua, 1e-9) * 100 >= args.min_pct):
            print("%-52s %-34s %9.1f %9.1f %+8.1f" % (t[:52], (fa if fa == fb else fa + " -> " + fb)[:34], ua, ub, d))
    ta, tb = sum(u for u, _ in A.values()), sum(u for u, _ in B.values())
    print("total in-scope us/step: A %.1f  B %.1f  (%+.1f)" % (ta, tb, tb - ta))


if __name__ == "__main__":
    main()
