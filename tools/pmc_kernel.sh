#!/bin/bash
# SQ counters of the dominant kernel (one rocprofv3 --pmc pass, no tracing).  usage: [ROOFLINE_KERNEL=wgrad] tools/pmc_kernel.sh <tag> <math> <batch> COUNTER...
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; MATH=$2; B=$3; shift 3
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT -o p -- python3 $R/bench.py --roofline-only ${ROOFLINE_KERNEL:+--roofline-kernel $ROOFLINE_KERNEL} --math $MATH --batch $B > $OUT/run.log 2>&1
echo "rc=$?"
cd $R && python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if any(k in r["Kernel_Name"] for k in ("conv_bf16_v2_kernel", "conv_bf16_kernel", "igemm_fwd_kernel", "igemm_wgrad")):
        a = acc[(r["Kernel_Name"][:60], r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print("%-62s %-28s avg %14.0f  (%d launches)" % (k, c, v / n, n))
PY
