"""Condense tools/profile_roofline.sh's three rocprofv3 passes into the numbers bench.py reports beside its live timing:
average kernel duration (kernel-trace pass) and HBM traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes
(MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced
reads).  Prints a JSON fragment for profiles/roofline_profiled.json and copies the small CSVs next to it.
FETCH_SIZE is doubled only for kernels that read whole 128-byte lines (the fp32 gather); the bf16 convolution's 64-byte
pieces and the bf16 weight gradient's pixel-major pieces are counted exactly (profiles/r03_fetch_calibration.md).
usage: python tools/roofline_profiled.py <gpurun_out/roofline_tag> <tag> [kernel-name-substring] [fetch scale]"""
import collections, csv, glob, json, os, shutil, sys

out, tag = sys.argv[1], sys.argv[2]
KEY = (sys.argv[3],) if len(sys.argv) > 3 else ("conv_bf16_v2_kernel", "conv_bf16_kernel", "igemm_fwd_kernel", "igemm_fwd_split_kernel")
SCALE = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0


def dominant(rows, name_col):
    acc = collections.defaultdict(list)
    for r in rows:
        n = r[name_col]
        if any(k in n for k in KEY):
            acc[n].append(r)
    return max(acc.items(), key=lambda kv: len(kv[1])) if acc else (None, [])


stats = list(csv.DictReader(open(glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)[0])))
name, rows = dominant(stats, "Name")
avg_ns = float(rows[0]["AverageNs"]); calls = int(rows[0]["Calls"])
res = {"kernel": name, "calls": calls, "avg_ns": avg_ns}
for what in ("fetch", "write"):
    f = glob.glob(out + "/%s/**/*counter_collection.csv" % what, recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Kernel_Name"] == name]
    res[what + "_kib_avg"] = sum(vals) / len(vals)
    res[what + "_launches"] = len(vals)
res["fetch_scale"] = SCALE
res["traffic_bytes"] = (SCALE * res["fetch_kib_avg"] + res["write_kib_avg"]) * 1024
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
res["sources_sha16"] = bench.sources_sha16()
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "roofline_%s_summary" % tag)
os.makedirs(dst, exist_ok=True)
shutil.copy(glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)[0], os.path.join(dst, "kernel_stats.csv"))
for what in ("fetch", "write"):
    f = glob.glob(out + "/%s/**/*counter_collection.csv" % what, recursive=True)[0]
    # keep only the dominant kernel's rows (the full file lists every launch of every kernel)
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"] == name]
    with open(os.path.join(dst, "%s_counter.csv" % what), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size", "Workgroup_Size"])
        w.writeheader()
        for r in rows:
            w.writerow({k: r.get(k, "") for k in w.fieldnames})
json.dump(res, open(os.path.join(dst, "summary.json"), "w"), indent=1)
print(json.dumps(res))
