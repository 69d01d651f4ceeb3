"""Kernel totals of the generator's two single-stream pieces (forward; backward + Adam + EMA) in the LAST step of a rocprofv3
--kernel-trace CSV of bench.py (dev tool).  The main stream is the queue of the ema kernel."""
import collections, csv, re, sys
rows = []
with open(sys.argv[1]) as fp:
    for r in csv.DictReader(fp):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"])))
rows.sort()
ema = [i for i, r in enumerate(rows) if "ema_kernel" in r[2]]
seg = rows[ema[-2] + 1:ema[-1] + 1]
mainq = rows[ema[-1]][3]
main = [r for r in seg if r[3] == mainq]
others = [r for r in seg if r[3] != mainq]
t_first_other, t_last_other = min(r[0] for r in others), max(r[1] for r in others)
fwd = [r for r in main if r[1] <= t_first_other + 1]
bwd = [r for r in main if r[0] >= t_last_other - 1]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*$", "", n)[:60]


for name, part in (("generator forward", fwd), ("generator backward + Adam + EMA", bwd)):
    by = collections.OrderedDict()
    for s, e, n, q in part:
        d = by.setdefault(short(n), [0, 0.0]); d[0] += 1; d[1] += (e - s) / 1e3
    tot = sum(v[1] for v in by.values())
    wall = (part[-1][1] - part[0][0]) / 1e3 if part else 0
    print("%s: %d launches, kernel time %.0f us, wall %.0f us" % (name, len(part), tot, wall))
    for k, (c, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:22]:
        print("   %8.1f us %3d x %7.1f  %s" % (us, c, us / c, k))
