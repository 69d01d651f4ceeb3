"""Per-kernel totals of the LAST train step of two rocprofv3 --kernel-trace CSVs side by side (dev tool): e.g. the eager
step against the same step replayed from its recording.  usage: trace_compare.py a.csv b.csv"""
import collections, csv, re, sys


def load(path):
    rows = []
    with open(path) as fp:
        for r in csv.DictReader(fp):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"])))
    rows.sort()
    ema = [i for i, r in enumerate(rows) if "ema_kernel" in r[2]]
    seg = rows[ema[-2] + 1:ema[-1] + 1]
    wall = (rows[ema[-1]][1] - rows[ema[-2]][1]) / 1e6
    by = collections.OrderedDict()
    for s, e, n, q in seg:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(.*$", "", n)[:64]
        d = by.setdefault(n, [0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
    queues = collections.Counter(q for _, _, _, q in seg)
    return wall, len(seg), by, queues


wa, na, a, qa = load(sys.argv[1])
wb, nb, b, qb = load(sys.argv[2])
print("A: wall %.2f ms, %d launches, queues %s" % (wa, na, dict(qa)))
print("B: wall %.2f ms, %d launches, queues %s" % (wb, nb, dict(qb)))
print("kernel time A %.2f ms, B %.2f ms" % (sum(v[1] for v in a.values()) / 1e3, sum(v[1] for v in b.values()) / 1e3))
names = sorted(set(a) | set(b), key=lambda k: -(a.get(k, [0, 0])[1] + b.get(k, [0, 0])[1]))
print("%-64s %5s %9s | %5s %9s" % ("kernel", "nA", "usA", "nB", "usB"))
for k in names[:45]:
    ca, ua = a.get(k, [0, 0.0]); cb, ub = b.get(k, [0, 0.0])
    print("%-64s %5d %9.1f | %5d %9.1f" % (k, ca, ua, cb, ub))
