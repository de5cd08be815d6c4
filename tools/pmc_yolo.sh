#!/bin/bash
# On the GPU box: hardware counters of the detector's kernels in one mode (tools/bench_yolo.py under rocprofv3 --pmc, one pass per counter group).
#   bash tools/pmc_yolo.sh <tag> <mode> "<counters of pass 1>" ["<counters of pass 2>" ...]   -> gpurun_out/<tag>_p<i>/
set -e
TAG=$1; MODE=$2; shift 2
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
i=1
for grp in "$@"; do
  rocprofv3 --pmc $grp --output-format csv -d $REPO/gpurun_out/${TAG}_p$i -- python3 $REPO/tools/bench_yolo.py 32 $MODE > $REPO/gpurun_out/${TAG}_p$i.log 2>&1
  echo pass $i done
  i=$((i+1))
done
