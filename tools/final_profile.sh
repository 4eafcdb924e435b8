#!/bin/bash
# Round-end measurement set on the GPU box (run from the repo root through gpurun):
#   kernel trace + stats of the bench command, the two PMC passes for HBM traffic, the bench line itself.
# usage: bash tools/final_profile.sh TAG        -> gpurun_out/TAG/*
set -o pipefail
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="bench.py --steps 10 --warmup 3 --cpu-sample 0 --e2e-reads 0"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $CMD > $OUT/stats.json 2> $OUT/stats.err || exit 1
PMC="bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e-reads 0"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $PMC > $OUT/fetch.json 2> $OUT/fetch.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $PMC > $OUT/write.json 2> $OUT/write.err || exit 1
python3 tools/pmc_traffic.py $OUT/fetch/f_counter_collection.csv $OUT/write/w_counter_collection.csv $OUT/traffic.json > $OUT/pmc_traffic.md || exit 1
timeout -k 10 600 python3 bench.py --traffic-json $OUT/traffic.json > $OUT/bench_line.json 2> $OUT/bench.err || exit 1
tail -c 600 $OUT/bench_line.json
