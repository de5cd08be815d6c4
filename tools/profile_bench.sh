#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + the PMC passes of one bench.py workload (default: the headline).
#   bash tools/profile_bench.sh <tag> [workload] [sq]   -> gpurun_out/<tag>_kt, <tag>_pmc_fetch, <tag>_pmc_write [, <tag>_pmc_sq]
# Then, back in the container:
#   python tools/summarize_prof.py <tag> gpurun_out/<tag>_kt gpurun_out/<tag>_pmc_fetch gpurun_out/<tag>_pmc_write 512 <workload> [gpurun_out/<tag>_pmc_sq]
# (512 = images per front-end launch at the default 256 stereo lanes).  Counters are collected in their own runs, never with a trace;
# python3 itself follows `--` (no env / bash -c hop under the profiler).
set -e
TAG=${1:-prof}
WL=${2:-stereo-yolo}
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/${TAG}_kt -- python3 $REPO/bench.py --workload $WL --extra none --cpu-budget 0 > $REPO/gpurun_out/${TAG}_kt.log 2>&1
echo kernel-trace done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/${TAG}_pmc_fetch -- python3 $REPO/bench.py --workload $WL --steps 2 --warmup 0 --extra none --cpu-budget 0 --no-profile > $REPO/gpurun_out/${TAG}_pmc_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/${TAG}_pmc_write -- python3 $REPO/bench.py --workload $WL --steps 2 --warmup 0 --extra none --cpu-budget 0 --no-profile > $REPO/gpurun_out/${TAG}_pmc_write.log 2>&1
echo write done
if [ "$3" = "sq" ]; then
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $REPO/gpurun_out/${TAG}_pmc_sq -- python3 $REPO/bench.py --workload $WL --steps 2 --warmup 0 --extra none --cpu-budget 0 --no-profile > $REPO/gpurun_out/${TAG}_pmc_sq.log 2>&1
echo sq done
fi
echo done
