"""Timeline analysis of a rocprofv3 --kernel-trace CSV of bench.py (dev tool): per step, wall time, time with no
kernel running (idle gaps), time with exactly one queue busy, and the kernels that run alone the longest."""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as fp:
    for r in csv.DictReader(fp):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"])))
rows.sort()
# step boundaries: the EMA kernel ends each step
ema = [i for i, r in enumerate(rows) if "ema_kernel" in r[2]]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ema = ema[-(nsteps + 1):]
for a, b in zip(ema[:-1], ema[1:]):
    seg = rows[a + 1:b + 1]
    t0, t1 = rows[a][1], rows[b][1]
    ev = []
    for s, e, n, q in seg:
        ev.append((s, 1, n))
        ev.append((e, -1, n))
    ev.sort()
    active = 0
    last = t0
    idle = solo = multi = 0
    solo_by = collections.Counter()
    cur = {}
    for t, d, n in ev:
        dt = t - last
        if active == 0:
            idle += dt
        elif active == 1:
            solo += dt
            solo_by[next(iter(cur))[:60]] += dt
        else:
            multi += dt
        last = t
        if d == 1:
            cur[n] = cur.get(n, 0) + 1
        else:
            cur[n] -= 1
            if cur[n] == 0:
                del cur[n]
        active += d
    print("step: wall %.2f ms  idle %.2f  one kernel %.2f  overlapped %.2f  launches %d" % (
        (t1 - t0) / 1e6, idle / 1e6, solo / 1e6, multi / 1e6, len(seg)))
print("kernels running alone (last step), ms:")
for n, v in solo_by.most_common(14):
    print("  %6.2f  %s" % (v / 1e6, n))
