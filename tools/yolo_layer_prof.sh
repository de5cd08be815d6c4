#!/bin/bash
# On the GPU box: per-layer table of the detector in mode f32w from a kernel trace.  usage: bash tools/yolo_layer_prof.sh <tag> [mode=f32w]
set -e
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format rocpd -d $REPO/gpurun_out/$1 -- python3 $REPO/tools/bench_yolo.py 128 ${2:-f32w} > $REPO/gpurun_out/$1.log 2>&1
cd $REPO
DB=$(find gpurun_out/$1 -name "*.db" | head -1)
python3 tools/yolo_layer_table.py $DB 128 > gpurun_out/$1_layers.txt
