"""Sequential listing of the kernels of the LAST train step in a rocprofv3 --kernel-trace CSV of bench.py (dev tool):
start offset (us), queue, duration, gap to the previous kernel end on ANY queue, kernel name.  Optional second
argument: 'g' lists only the G update (after the last discriminator Adam), 'summary' groups the G update by kernel."""
import collections
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as fp:
    for r in csv.DictReader(fp):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"])))
rows.sort()
mode = sys.argv[2] if len(sys.argv) > 2 else "all"
ema = [i for i, r in enumerate(rows) if "ema_kernel" in r[2]]
a, b = ema[-2], ema[-1]
seg = rows[a + 1:b + 1]
t0 = rows[a][1]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:70]


adams = [i for i, r in enumerate(seg) if "adam_kernel" in r[2]]
g_start = adams[-2] + 1 if len(adams) >= 2 else 0   # after the last D adam (the last adam is G's)
if mode in ("g", "summary"):
    # the G update starts with the D passes on fake images, which begin before the last D Adam; use the first kernel after
    # the third-last adam as a conservative start
    seg = seg[g_start:]
print("step wall %.2f ms, %d launches listed" % ((rows[b][1] - t0) / 1e6, len(seg)))
if mode == "summary":
    by = collections.OrderedDict()
    for s, e, n, q in seg:
        k = short(n)
        d = by.setdefault(k, [0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
    tot = sum(v[1] for v in by.values())
    print("kernel time %.2f ms over wall %.2f ms" % (tot / 1e3, (seg[-1][1] - seg[0][0]) / 1e6))
    for k, (c, us) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print("%8.1f us %4d x %7.1f  %s" % (us, c, us / c, k))
else:
    last_end = seg[0][0]
    for s, e, n, q in seg:
        print("%9.1f q%-2d %8.1f us  gap %7.1f  %s" % ((s - t0) / 1e3, q, (e - s) / 1e3, (s - last_end) / 1e3, short(n)))
        last_end = max(last_end, e)
