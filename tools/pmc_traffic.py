#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs
as MI355X_MICROARCH.md 'rocprofv3 PMC slots' requires).

Units / corrections (MI355X_MICROARCH.md section HBM): both counters are in KiB; on gfx950
FETCH_SIZE tallies 128-B requests at 64 B, i.e. reports exactly 1/2 of a wide coalesced read
stream, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE reads exactly.  The factor is
calibrated for 16-B-per-lane streams only; for this path's mix of 8-B coalesced reads and
random 16-B slot accesses it is applied as prescribed and flagged as uncalibrated.
Writes JSON {kernel: bytes_per_launch}.
"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main(fetch_csv, write_csv, out_json):
    f, nf = per_kernel(fetch_csv, "FETCH_SIZE")
    w, _ = per_kernel(write_csv, "WRITE_SIZE")
    out = {}
    rows = []
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, 0) + w.get(k, 0))):
        rd = 2.0 * f.get(k, 0.0) * 1024.0
        wr = w.get(k, 0.0) * 1024.0
        out[k] = round(rd + wr)
        rows.append((k, nf.get(k, 0), f.get(k, 0.0), w.get(k, 0.0), rd + wr))
    json.dump(out, open(out_json, "w"), indent=1)
    print("| kernel | launches | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM bytes/launch (2*F+W) |")
    print("|---|---:|---:|---:|---:|")
    for k, n, a, b, t in rows[:24]:
        print("| `%s` | %d | %.1f | %.1f | %.3e |" % (k[:70], n, a, b, t))
    # whole pass: every launch of every kernel, divided by the passes of the run (the torch fill kernels
    # of the bench's own buffers run once, not per pass: left out)
    passes = max([nf.get(k, 0) for k in nf if k.startswith(("k_dedup_lds", "k_dedup_rec"))] or [1])
    own = [(k, n, a, b) for k, n, a, b, _ in rows if not k.startswith("at::")]
    rd = sum(2.0 * a * 1024.0 * n for _, n, a, _ in own) / passes
    wr = sum(b * 1024.0 * n for _, n, _, b in own) / passes
    print("\nWhole pass (all kernels of one hot-path pass, 2*FETCH_SIZE + WRITE_SIZE; %d passes in the run): "
          "**%.2f GB** (reads %.2f GB, writes %.2f GB)" % (passes, (rd + wr) / 1e9, rd / 1e9, wr / 1e9))


if __name__ == "__main__":
    main(*sys.argv[1:4])
