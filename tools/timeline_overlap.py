#!/usr/bin/env python3
"""Is the association on the critical path?  From a rocprofv3 *_kernel_trace.csv of bench.py (tools/profile_bench.sh keeps it for this):
the wall time of the steady part of the run is split into
  conv        time in which at least one convolution kernel (detector or ReID network: the chip-filling work) is executing
  assoc_only  time in which tracker kernels (trk_* / lsap / kalman / cost) run and NO convolution kernel does - association EXPOSED on the
              critical path: the matrix cores wait for the tracker
  other_only  neither of the two but some other kernel (resize, NMS, decode, copies)
  idle        no kernel at all
and the tracker kernels' own busy time is reported with the share of it that runs UNDER convolution kernels (hidden).

    python tools/timeline_overlap.py <kernel_trace.csv> [from = 0.3] [to = 1.0]  > profiles/rNN_<cfg>_timeline.txt
(fractions of the span between the first and the last tracker kernel, i.e. of the pipeline steps of the run: tools/profile_bench.sh
TIMELINE=1 traces `bench.py --no-roofline --no-extras --steps 60` and cuts the window inside the timed steps).  Under the profiler every dispatch carries tracing overhead
(the run is ~15 % slower than untraced), which shows up as "no kernel at all"; the question the table answers is what runs BESIDE what."""
import csv
import sys


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(iv):
    return sum(b - a for a, b in iv)


def intersect(x, y):
    i = j = 0
    out = []
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if a < b:
            out.append([a, b])
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return out


def main(path, skip=0.3, upto=1.0):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name") or r.get("Name")))
    is_trk = lambda n: any(k in n for k in ("trk_", "lsap", "kf_", "kalman", "cost_kernel", "normalize_rows", "gallery"))
    # the reference span = first .. last tracker kernel: the pipeline steps of the run (plan-time autotune launches and the host-side
    # set-up between them and the first step carry no tracker kernel); `skip` drops the schedule trial and warm-up steps at its front
    steps = [r for r in rows if is_trk(r[2])]
    t_lo, t_hi = min(r[0] for r in steps), max(r[1] for r in steps)
    cut = t_lo + int((t_hi - t_lo) * skip)
    end = t_lo + int((t_hi - t_lo) * upto)
    rows = [r for r in rows if r[0] >= cut and r[1] <= end]
    t_lo, t_hi = min(r[0] for r in rows), max(r[1] for r in rows)
    is_conv = lambda n: "conv" in n
    conv = union([(a, b) for a, b, n in rows if is_conv(n)])
    trk = union([(a, b) for a, b, n in rows if is_trk(n)])
    other = union([(a, b) for a, b, n in rows if not is_conv(n) and not is_trk(n)])
    anyk = union([(a, b) for a, b, _ in rows])
    wall = t_hi - t_lo
    trk_under = length(intersect(trk, conv))
    trk_only = length(trk) - trk_under
    oth_only = length(other) - length(intersect(other, union(conv + trk)))
    n_trk = sum(1 for _, _, n in rows if is_trk(n))
    print(f"trace window            {wall / 1e6:10.2f} ms  ({len(rows)} kernel launches, {skip:.0%} .. {upto:.0%} of the span of pipeline steps)")
    print(f"conv kernels active     {length(conv) / 1e6:10.2f} ms  {length(conv) / wall:7.1%} of the wall time")
    print(f"tracker kernels active  {length(trk) / 1e6:10.2f} ms  {length(trk) / wall:7.1%}   ({n_trk} launches)")
    print(f"  under conv kernels    {trk_under / 1e6:10.2f} ms  {trk_under / max(length(trk), 1):7.1%} of the tracker's busy time is hidden")
    print(f"  EXPOSED (no conv)     {trk_only / 1e6:10.2f} ms  {trk_only / wall:7.1%} of the wall time")
    print(f"other kernels only      {oth_only / 1e6:10.2f} ms  {oth_only / wall:7.1%}")
    print(f"no kernel at all        {(wall - length(anyk)) / 1e6:10.2f} ms  {(wall - length(anyk)) / wall:7.1%}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.3, float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
