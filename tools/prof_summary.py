#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats CSV (…_kernel_stats.csv) into a short markdown table."""
import csv
import sys


def main(path, title, steps):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("# %s\n" % title)
    print("source: `%s`; total kernel time %.3f ms over %d hot-path passes (%.3f ms/pass)\n"
          % (path, tot / 1e6, steps, tot / 1e6 / steps))
    print("| kernel | calls | avg us | total ms | % |")
    print("|---|---:|---:|---:|---:|")
    for r in rows[:28]:
        name = r["Name"]
        if "rocprim" in name:
            # keep the innermost kernel name
            i = name.find("detail::", name.find("trampoline_kernel") + 1)
            name = "rocprim::" + name[i + 8:i + 70] if i > 0 else name[:70]
        print("| `%s` | %s | %.1f | %.3f | %.2f |" % (name[:80].replace("|", "/"), r["Calls"],
              float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
