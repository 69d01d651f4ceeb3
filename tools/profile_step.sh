#!/bin/bash
# rocprofv3 kernel statistics of a bench.py run (GPU box).  usage: tools/profile_step.sh <outdir-under-gpurun_out> <steps+warmup+1> "<title>" <bench args...>
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; NSTEPS=$2; TITLE=$3; shift 3
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $R/bench.py "$@" > $OUT/run.log 2>&1
echo "rocprofv3 rc=$?"
tail -1 $OUT/run.log | cut -c1-400
cd $R
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 tools/profile_summary.py "$F" "$NSTEPS" "$TITLE" "rocprofv3 --kernel-trace --stats -- python3 bench.py $*" > $OUT/summary.md
cp "$F" $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -size +40M -delete
head -60 $OUT/summary.md
