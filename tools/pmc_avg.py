"""Average of one PMC counter per kernel from a rocprofv3 --pmc run directory (dev tool)."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][:70], r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"])
    acc[k][1] += 1
for (k, c), (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:6]:
    print("%-70s %-12s calls %4d  avg %.1f" % (k, c, n, v / n))
