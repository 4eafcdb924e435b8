#!/bin/bash
# SQ counters of the hot kernels: one rocprofv3 --pmc pass per counter group (kernel trace only), summarised per kernel.
# usage (through gpurun, from the repo root): bash tools/pmc_sq.sh TAG [kernel-name-substring ...]
set -o pipefail
TAG=${1:-sq}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e-reads 0 --no-verify"
i=0
for GROUP in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" \
             "SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM" "SQ_LDS_ATOMIC_RETURN SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d $OUT/g$i -o c -- python3 $CMD > $OUT/g$i.json 2> $OUT/g$i.err || echo "group $i failed: $GROUP" >> $OUT/failed.txt
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]; names = sys.argv[2:] or ["k_dedup_rec", "k_unperm_bins8", "k_p8_scatter1", "k_p8_scatter2", "k_group_fine", "k_pairs_append"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/c_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for n in names:
            if n in k:
                acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/sq_summary.md", "w") as fh:
    for n in names:
        fh.write("## %s\n" % n)
        for c, v in sorted(acc[n].items()):
            fh.write("%-28s mean %.4g over %d launches\n" % (c, sum(v) / len(v), len(v)))
print(open(out + "/sq_summary.md").read())
PY
