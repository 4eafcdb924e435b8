#!/bin/bash
# Per-phase latencies inside the hot kernels (common.hip.h, HUMID_PHASE_CLOCKS): builds an instrumented copy of the
# library, runs the bench with it, restores the product library.  Run through gpurun from the repo root:
#   bash tools/phase_clocks.sh TAG      -> gpurun_out/TAG/phase_clocks.txt
# (the instrumented library is built HERE, before gpurun sends the tree: tools/phase_clocks.sh build)
set -o pipefail
if [ "$1" = build ]; then
  mkdir -p tools/dbg/tmp
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DHUMID_PHASE_CLOCKS -o tools/dbg/tmp/libhumid_clk.so humid_amd/csrc/humid_hip.hip humid_amd/csrc/humid_exchange.hip humid_amd/csrc/shm.cpp
  exit $?
fi
TAG=${1:-clk}
mkdir -p gpurun_out/$TAG
cp humid_amd/libhumid_hip.so /tmp/libhumid_product.so
cp tools/dbg/tmp/libhumid_clk.so humid_amd/libhumid_hip.so
python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 --e2e-reads 0 --no-verify > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
cp /tmp/libhumid_product.so humid_amd/libhumid_hip.so
grep "phase clocks" gpurun_out/$TAG/bench.err > gpurun_out/$TAG/phase_clocks.txt
cat gpurun_out/$TAG/phase_clocks.txt
