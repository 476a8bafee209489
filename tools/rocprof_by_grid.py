#!/usr/bin/env python3
"""Per (conv tile variant, grid size) averages from a rocprofv3 *_kernel_trace.csv: the detector's launches of a kernel and the
ReID network's (and any autotune launch) differ in their grid, so this separates what the pooled *_kernel_stats.csv mixes.

    python tools/rocprof_by_grid.py <kernel_trace.csv> [variant substring]  > profiles/rNN_<cfg>_conv_by_grid.txt
Compare `avg us` of the detector rows with bench.py's roofline.avg_launch_us (event timed, detector stream only)."""
import csv
import sys
from collections import defaultdict

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from rocprof_agg import key_of  # noqa: E402


def main(path, only=None):
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or r.get("Name")
            k = key_of(name if name.startswith("void") else "void " + name)
            if k is None or (only and only not in k):
                continue
            grid = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
            wg = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 1)
            a = agg[(k, grid // max(wg, 1))]
            a[0] += 1
            a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    print(f"{'kernel (tile variant)':52s} {'workgroups':>10s} {'calls':>7s} {'avg us':>9s} {'total ms':>9s}")
    for (k, g), (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:52s} {g:10d} {n:7d} {ns / n / 1e3:9.1f} {ns / 1e6:9.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
