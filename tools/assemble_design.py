"""DESIGN.md = docs/design_parts/*.md in order (the numbers table of section 5 is filled from profiles/r06_bench_*.json)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "docs", "design_parts")


def line(name):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    rows = [l for l in open(path) if l.startswith("{")]
    return json.loads(rows[-1]) if rows else None


def table():
    out = ["| config (frames per step) | `value` (schedule) | with upload | other schedule | f32 math (`roofline_f32.frac`) | half | frame by frame / +1 | `VideoDetector.detect` | dominant kernel `frac` in the pipeline / at the sustained sclk | `mfma_busy` / HBM bytes per launch (PMC) | all conv `frac` / of attainable |",
           "|---|---|---|---|---|---|---|---|---|---|---|"]
    for cfg in ("cfg2", "cfg3", "cfg5"):
        d = line(f"r06_bench_{cfg}.json")
        if d is None:
            out.append(f"| {cfg} | (not measured yet) | | | | | | | | | |")
            continue
        r = d.get("roofline") or {}
        a = r.get("all_conv_kernels") or {}
        f32 = d.get("roofline_f32") or {}
        oth = d.get("value_other_schedule") or {}
        out.append("| {} ({}) | **{}** ({}) | {} | {} {} | {} ({}) | {} | {} / {} | {} | {} {} / {} at {} GHz | {} / {} MB | {} / {} |".format(
            cfg, d["config"]["frames_per_step"], d["value"], d["config"]["schedule"], d.get("value_with_upload"), oth.get("value"), oth.get("schedule", ""),
            d.get("value_f32_math"), f32.get("frac"), d.get("value_half_mode"), d.get("value_frame_by_frame"), d.get("value_frame_by_frame_lookahead1"),
            d.get("value_video_detector"), (r.get("kernel") or "").replace("conv3x3_f16x3_", ""), r.get("frac"), r.get("frac_at_sustained_clock"),
            r.get("sustained_clock_ghz"), r.get("mfma_busy"), None if not r.get("traffic") else round(r["traffic"] / 1e6), a.get("frac"), a.get("frac_of_attainable")))
    return "\n".join(out)


def sustained():
    d = line("r06_bench_cfg2_sustained.json")
    if d is None:
        return "(not measured yet)"
    r = d.get("roofline") or {}
    return ("`bench.py --steps {} --warmup 5` = {} frames in {:.1f} s: **{} frames/s**, dominant kernel `frac` {} at {} GHz sclk "
            "(`profiles/r06_bench_cfg2_sustained.json`) - the 20-step headline does not ride boost headroom.").format(
        d["steps"], d["steps"] * d["config"]["frames_per_step"], d["steps"] * d["ms_per_step"] / 1e3, d["value"], r.get("frac"), r.get("sustained_clock_ghz"))


parts = ["design_head.md", "design_s1.md", "design_s2.md", "design_s3.md", "design_s4.md", "design_s5.md", "design_s6.md", "design_s7.md"]
text = "".join(open(os.path.join(P, p)).read().rstrip() + "\n\n" for p in parts)
text = text.replace("ROUND6_TABLE", table()).replace("ROUND6_SUSTAINED", sustained())
open(os.path.join(ROOT, "DESIGN.md"), "w").write(text.rstrip() + "\n")
print("DESIGN.md", len(text.splitlines()), "lines")
