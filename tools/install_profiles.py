"""Build container: copy what tools/collect_profiles.sh <tag> left under gpurun_out/ into profiles/ and rewrite
profiles/roofline_profiled.json from the three summary.json files (dev tool).  usage: python tools/install_profiles.py r03
The bf16 weight gradient reads whole 128 / 256-byte pixels of both operands: its FETCH_SIZE is doubled like the fp32 gather's
(profiles/r03_fetch_calibration.md); the bf16 convolution's 64-byte pieces are counted exactly."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
WHY = {1.0: "x1: 64-byte pieces are counted exactly (profiles/r03_fetch_calibration.md)",
       2.0: "x2: whole 128-byte lines are counted at half (profiles/r03_fetch_calibration.md)"}
out = {}
for key, scale in (("bf16_b48", 1.0), ("f32_b24", 2.0), ("bf16_wgrad_b48", 2.0)):
    src = os.path.join(G, "roofline_%s_%s_summary" % (tag, key))
    dst = os.path.join(P, "%s_roofline_%s" % (tag, key))
    os.makedirs(dst, exist_ok=True)
    s = json.load(open(os.path.join(src, "summary.json")))
    s["fetch_scale"] = scale
    s["traffic_bytes"] = (scale * s["fetch_kib_avg"] + s["write_kib_avg"]) * 1024
    for f in ("fetch_counter.csv", "write_counter.csv", "kernel_stats.csv"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    json.dump(s, open(os.path.join(dst, "summary.json"), "w"), indent=1)
    rel = "profiles/%s_roofline_%s" % (tag, key)
    out[key] = {"avg_ns": s["avg_ns"],
                "avg_source": "%s/kernel_stats.csv (rocprofv3 --kernel-trace --stats, %d launches)" % (rel, s["calls"]),
                "traffic_bytes": s["traffic_bytes"], "fetch_scale": scale,
                "traffic_source": "%s/{fetch,write}_counter.csv: (%g*FETCH_SIZE %.0f KiB + WRITE_SIZE %.0f KiB)*1024 per launch "
                                  "(separate --pmc passes; %s)" % (rel, scale, s["fetch_kib_avg"], s["write_kib_avg"], WHY[scale]),
                "kernel": s["kernel"], "sources_sha16": s["sources_sha16"]}
json.dump(out, open(os.path.join(P, "roofline_profiled.json"), "w"), indent=1)
COPY = {"bench_default_run.json": "%s_bench_default_run.json", "bench_bf16_b48.json": "%s_bench_bf16_b48.json",
        "conv16_layers.txt": "%s_bf16_conv_layers.txt", "layers_f32_b24.txt": "%s_f32_per_launch_table.txt",
        "timeline_bf16.txt": "%s_bf16_timeline.txt", "timeline_f32.txt": "%s_f32_timeline.txt",
        "image_layers.txt": "%s_image_layers.txt"}
for a, b in COPY.items():
    f = os.path.join(G, tag, a)
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, os.path.join(P, b % tag))
for step, name in (("step_f32", "f32_b24"), ("step_bf16", "bf16_b48")):
    d = os.path.join(G, tag, step)
    for f in os.listdir(d) if os.path.isdir(d) else ():
        if f.endswith("kernel_stats.csv"):
            shutil.copy(os.path.join(d, f), os.path.join(P, "%s_%s_step_kernel_stats.csv" % (tag, name)))
        if f.endswith(".md"):
            shutil.copy(os.path.join(d, f), os.path.join(P, "%s_%s_step_kernel_stats.md" % (tag, name)))
print(json.dumps({k: (v["kernel"][28:70], round(v["avg_ns"] / 1e3, 1), round(v["traffic_bytes"] / 1e6, 1), v["sources_sha16"])
                  for k, v in out.items()}, indent=1))
