#!/bin/bash
# FAST append variants (csrc/k_fast.h: SD_FAST_APPEND / SD_FAST_ADD1), timed back to back on one box.
#   in the build container:  bash tools/fast_append_ab.sh build     -> build/libsd_fast_{A,B,C,D}.so (build/ is git-ignored and travels with gpurun)
#   on the GPU box:          gpurun -- 'bash tools/fast_append_ab.sh'
if [ "$1" = "build" ]; then
  set -e
  mkdir -p build
  python -c "import __graft_entry__ as g; g.build()"
  F="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude"
  hipcc $F "-DSD_FAST_ADD1(p)=sd_lds_add_rtn((p),1)" -c -o build/a.o slam-dynamic_amd/csrc/sd_api.hip
  hipcc $F -DSD_FAST_APPEND=1 "-DSD_FAST_ADD1(p)=sd_lds_add_rtn((p),1)" -c -o build/b.o slam-dynamic_amd/csrc/sd_api.hip
  hipcc $F -DSD_FAST_APPEND=1 -c -o build/c.o slam-dynamic_amd/csrc/sd_api.hip
  hipcc $F -c -o build/d.o slam-dynamic_amd/csrc/sd_api.hip
  for v in a:A b:B c:C d:D; do hipcc --offload-arch=gfx950 -shared -fPIC -o build/libsd_fast_${v#*:}.so build/${v%:*}.o slam-dynamic_amd/lib/sd_yolo_api.o; done
  rm -f build/?.o
  exit 0
fi
set -e
mkdir -p gpurun_out
for v in A B C D A B C D; do
  L=$PWD/build/libsd_fast_$v.so
  SD_FRONTEND_LIB=$L timeout -k 10 120 python -m pytest tests/test_gpu_extract.py -m gpu -x -q > gpurun_out/fast_$v.t.log 2>&1 || { echo "variant $v FAILED parity"; tail -5 gpurun_out/fast_$v.t.log; exit 1; }
  SD_FRONTEND_LIB=$L timeout -k 10 200 python bench.py --workload stereo --extra none --cpu-budget 0 --detail fast_$v.json > gpurun_out/fast_$v.log 2>&1
  python3 - <<PY
import json
j=json.load(open("fast_$v.json")); r=j["roofline"]
print("$v", j["value"], j["ms_per_step"], "k_fast_cells", r["kernels_ms_per_step"]["k_fast_cells"])
PY
done
