#!/usr/bin/env python3
"""Aggregate a rocprofv3 *_kernel_stats.csv by conv tile variant so that it can be compared with bench.py's
`conv_variants` / `roofline.avg_launch_us` (bench pools the activation / residual / input-format template
arguments of one tile shape; rocprof lists every instantiation separately).

    python tools/rocprof_agg.py profiles/r01_bench_cfg2_kernel_stats.csv
"""
import csv
import re
import sys
from collections import OrderedDict


def key_of(name):
    m = re.match(r"void yds::(?:\(anonymous namespace\)::)?(conv3x3_f16x3_win16|conv3x3_f16x3_win2|conv3x3_f16x3_win|conv_igemm_f16x3_dma|conv_igemm_f16x3|conv_igemm_f32|conv3x3_rgb_direct|conv3x3_rgb_pool_mfma|conv3x3_rgb_pool|conv_stem2_f16x3|conv_block1_f16x3)<([^>]*)>", name)
    if not m:
        return None
    kind, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
    if kind == "conv_igemm_f16x3":
        return f"{kind}<{args[0]},{args[1]}>"
    if kind == "conv_igemm_f16x3_dma":
        return f"{kind}<{args[0]},{args[1]},{args[2]}x{args[3]},{args[4]}>"
    if kind == "conv3x3_f16x3_win2":
        return f"{kind}<128,128,2x2>"
    if kind == "conv3x3_f16x3_win16":       # the default-arithmetic form of the window kernel (16x16x32 MFMA): same tile variant name as bench.py's
        return f"conv3x3_f16x3_win<{args[0]},{args[1]},{args[2]}x{args[3]}>"       # template <BM, BN, WM, WN, ...>
    if kind == "conv3x3_f16x3_win":
        return f"{kind}<256,{args[0]},{args[1]}x{args[2]}>"
    if kind == "conv_igemm_f32":
        return f"{kind}<{','.join(args[:5])}>"
    return kind


def main(path):
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    agg = OrderedDict()
    other = 0.0
    for r in rows:
        k = key_of(r["Name"])
        if k is None:
            other += float(r["TotalDurationNs"])
            continue
        a = agg.setdefault(k, [0, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
    print(f"{'kernel (tile variant, all epilogue instantiations pooled)':60s} {'calls':>8s} {'avg us':>10s} {'share':>7s}")
    for k, (calls, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:60s} {calls:8d} {ns / calls / 1e3:10.1f} {100 * ns / total:6.2f}%")
    print(f"{'all non-conv kernels':60s} {'':8s} {'':10s} {100 * other / total:6.2f}%")
    print(f"total kernel time {total / 1e6:.1f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
