"""Long-stream parity AT the benchmarked shape (VERDICT r4 'next' #1): 256 / 128 / 96 consecutive 1080p frames of the cfg2 /
cfg3 / cfg5 streams - with synth.PersonScene's long occlusion windows, so that confirmed tracks die of max_age = 30
(reference deep_sort/sort/track.py:146-152), persons return under new ids, tracks are re-identified after 12-28 hidden
frames and the galleries run into nn_budget = 30 (nn_matching.py:152-155) - through the very Workload object bench.py
times (32 frames per step, next pass prefetched; the library's own schedule policy - which times both stream schedules on these very
steps and so switches mid-stream - and each schedule forced), against rows the REFERENCE produced on the same
frames (oracle/gen_golden.py gen_long_stream: the body of video_detect.py:134-157 with the real Extractor).

Exact: None-ness, row counts, track ids and classes of every frame; the track list (ids, states, time_since_update) after
every step.  Box columns: within one, and differing on < 0.5 % of the values (fp32 truncation next to an integer).

The fixture also carries how close the reference's own decisions came to their thresholds (MarginSpy).  The synthetic ReID
weights map every person to almost the same embedding (cosine costs ~3e-3 against the 0.3 threshold), so the appearance
stage is decided by cost differences of 1e-3 .. 1e-5: `margin_lsap_eps` records frames whose assignment flips under a
1e-4 perturbation of the cost matrix.  Bit-exact ids over these streams therefore hold the embeddings to ~1e-5."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu
B = 32


def _events(g):
    """(confirmed tracks that died, ids born after the warm-up, max galleries at budget) of the reference run."""
    n, ids, ptr, st = int(g["n_frames"]), g["ids"], g["ids_ptr"], g["state"]
    prev, deaths, births = {}, 0, 0
    for t in range(n):
        cur = dict(zip(ids[ptr[t]:ptr[t + 1]].tolist(), st[ptr[t]:ptr[t + 1]].tolist()))
        deaths += sum(1 for i, s in prev.items() if i not in cur and s == 2)
        births += sum(1 for i in cur if i not in prev and t > 3)
        prev = cur
    return deaths, births, int(g["at_budget"].max())


def _run(config, schedule):
    from yolo_deepsort_amd.workload import Workload
    g = golden(f"long_stream_{config}")
    n = int(g["n_frames"])
    assert n % B == 0
    wl = Workload(config, batch=B, n_distinct=n, long_occlude=n, pingpong=False)
    assert wl.order == list(range(n))
    assert np.array_equal(np.array([[p, a, b] for p, (a, b) in sorted(wl.scene.long_windows.items())]).reshape(-1, 3), g["windows"])
    wl.pipe.set_schedule(schedule)
    rows, ptr, iptr = g["out_rows"], g["out_ptr"], g["ids_ptr"]
    off = total = 0
    for i in range(n // B):
        outs = wl.step(i, prefetch=(i + 1 < n // B))
        st = wl.ds.tracker.state()            # after the step's last frame (a prefetched detector pass touches no tracker state)
        for b, o in enumerate(outs):
            t = i * B + b
            ref = rows[ptr[t]:ptr[t + 1]]
            if bool(g["none"][t]):
                assert o is None, t
                continue
            assert o is not None and o.shape == ref.shape, (t, None if o is None else o.shape, ref.shape)
            assert np.array_equal(o[:, 4:], ref[:, 4:]), (t, o[:, 4], ref[:, 4])          # track ids, classes
            d = np.abs(o[:, :4] - ref[:, :4])
            assert d.max(initial=0) <= 1, t
            off += int((d != 0).sum())
            total += d.size
        t = (i + 1) * B - 1
        sl = slice(iptr[t], iptr[t + 1])
        assert np.array_equal(st["ids"], g["ids"][sl]), t
        assert np.array_equal(st["state"], g["state"][sl]), t
        assert np.array_equal(st["tsu"], g["time_since_update"][sl]), t
    assert total > 0 and off / total < 5e-3, (off, total)
    return wl.pipe.last_schedule(), g, wl.pipe.schedule_trial()


@pytest.mark.parametrize("config,schedule,name", [("cfg2", None, "policy"), ("cfg2", -1, "two-stream"), ("cfg2", 256, "serialized"),
                                                  ("cfg3", None, "policy"), ("cfg3", -1, "two-stream"),
                                                  ("cfg5", None, "policy"), ("cfg5", 256, "serialized")])
def test_long_stream_ids_bit_exact_vs_reference(config, schedule, name):
    ran, g, trial = _run(config, schedule)
    if name == "policy":
        # the pipeline timed both schedules on the stream's own steps (5 serialized, then two-stream: pipeline.cpp Trial) - so this run
        # also crossed from one schedule to the other in the middle of the stream, with identical rows
        # (groups of 5 steady-state steps alternate serialized / two-stream; 20 steps decide - more than these streams have)
        n_steady = int(g["n_frames"]) // B - 1
        assert trial["decided"] is None, trial
        if n_steady >= 6:                        # cfg2: steps 0-4 serialized, 5-6 two-stream
            assert trial["serialized_s"] > 0 and ran == "two-stream", (trial, ran)
    else:
        assert ran == name and trial["decided"] is None
    deaths, births, at_budget = _events(g)
    # the stream really contains what the 32-frame fixtures cannot: deaths of confirmed tracks, re-births, full galleries
    assert deaths >= 3 and births >= 3 and at_budget >= 20, (deaths, births, at_budget)


def test_long_stream_margins_are_what_the_docstring_says():
    """The robustness numbers DESIGN.md quotes, read from the fixtures (no device work: kept with the test they explain)."""
    for config in ("cfg2", "cfg3", "cfg5"):
        g = golden(f"long_stream_{config}")
        cos, gate = g["margin_cos"], g["margin_gate"]
        assert np.nanmin(cos[np.isfinite(cos)]) > 0.25                     # every admissible cosine cost is far below 0.3 ...
        assert (g["margin_lsap_eps"] <= 1e-4).sum() >= 3                   # ... so frames exist whose assignment flips at 1e-4
        assert np.isfinite(gate).sum() > int(g["n_frames"]) // 2
