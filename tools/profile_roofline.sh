#!/bin/bash
# The dominant kernel alone under rocprofv3 (GPU box): kernel-trace average duration, then FETCH_SIZE and WRITE_SIZE in
# separate --pmc passes (MI355X_MICROARCH.md: the two TCC counters do not fit one pass).
# usage: tools/profile_roofline.sh <tag> <math> <batch> [conv|wgrad] [kernel-name-substring] [fetch scale]
#        -> gpurun_out/roofline_<tag>/{stats,fetch,write}/...
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; MATH=$2; B=$3; WHICH=${4:-conv}; KSUB=$5; FSCALE=$6
OUT=$R/gpurun_out/roofline_$TAG
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --roofline-only --math $MATH --batch $B --roofline-kernel $WHICH"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- $CMD > $OUT/stats.log 2>&1; echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1; echo "write rc=$?"
cd $R && python3 tools/roofline_profiled.py $OUT $TAG $KSUB $FSCALE
