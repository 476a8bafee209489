#!/usr/bin/env python3
"""Matrix-pipe utilisation per conv tile variant from ONE rocprofv3 --pmc pass of the bench command
(SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES + --kernel-trace, tools/profile_bench.sh MFMA=1).

    python tools/mfma_busy_from_pmc.py <counter_collection.csv> <kernel_trace.csv>  > mfma_busy.json
Per kernel and launch (means):
  mfma_busy      = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader cycles of the launch) - the fraction of matrix-pipe cycles
                   (summed over all SIMDs of the chip) in which an MFMA was executing; shader cycles = GRBM_GUI_ACTIVE / 8 XCDs
  clock_ghz      = shader cycles / kernel-trace duration (profiled launches are serialised by the profiler, so clocks run
                   higher than in back-to-back launches: compare `mfma_busy`, which is a cycle ratio, not the durations)
The MI355X has 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (profiles/r03_pmc_power.txt checked both)."""
import csv
import json
import sys
from collections import defaultdict

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from rocprof_agg import key_of  # noqa: E402


def main(counter_csv, trace_csv):
    per = defaultdict(lambda: defaultdict(float))      # dispatch id -> counter -> value
    name = {}
    for r in csv.DictReader(open(counter_csv)):
        d = r["Dispatch_Id"]
        per[d][r["Counter_Name"]] += float(r["Counter_Value"])
        name[d] = r["Kernel_Name"]
    dur = {}
    for r in csv.DictReader(open(trace_csv)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg = defaultdict(lambda: defaultdict(list))
    for d, c in per.items():
        k = key_of(name[d] if name[d].startswith("void") else "void " + name[d])
        if not k or "GRBM_GUI_ACTIVE" not in c or c["GRBM_GUI_ACTIVE"] <= 0:
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        agg[k]["mfma_busy"].append(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc))
        agg[k]["cycles"].append(cyc)
        if d in dur:
            agg[k]["us"].append(dur[d])
            agg[k]["ghz"].append(cyc / dur[d] / 1e3)
    mean = lambda v: sum(v) / len(v) if v else None
    out = {k: {"launches": len(v["mfma_busy"]), "mfma_busy": round(mean(v["mfma_busy"]), 4), "shader_cycles_per_launch": round(mean(v["cycles"])),
               "avg_us_profiled": None if not v["us"] else round(mean(v["us"]), 1), "clock_ghz_profiled": None if not v["ghz"] else round(mean(v["ghz"]), 3)}
           for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]["cycles"]))}
    json.dump({"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace over bench.py; "
                       "mfma_busy = MFMA_BUSY / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)", "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
