"""rocprofv3 `*_kernel_stats.csv` -> markdown table for profiles/ (dev tool).

usage: python tools/profile_summary.py <kernel_stats.csv> <steps_in_run> "<title>" "<command>" [note] > profiles/x.md
"""
import csv
import sys

path, steps, title, cmd = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
note = sys.argv[5] if len(sys.argv) > 5 else ""
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
ig = sum(float(r["TotalDurationNs"]) for r in rows if "igemm" in r["Name"] or "wgrad_k3" in r["Name"] or "small_n" in r["Name"])
calls = sum(int(r["Calls"]) for r in rows)
print("# %s\n" % title)
print("Command (MI355X box): `%s`\n" % cmd)
print("%d train steps in the run. Kernel time %.2f ms/step summed over all streams (the discriminator streams overlap, so "
      "this sum exceeds the wall time) over %d launches/step; convolution GEMM kernels %.2f ms/step (%.0f %%). %s\n"
      % (steps, tot / steps / 1e6, calls // steps, ig / steps / 1e6, 100 * ig / tot, note))
print("| % | calls/step | avg us | ms/step | kernel |\n|---|---|---|---|---|")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("| %.2f | %.1f | %.1f | %.2f | `%s` |" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]) / steps,
                                                 float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / steps / 1e6,
                                                 name[:100]))
