"""bench.py's `roofline` arithmetic (no GPU): a stage's bytes per frame are priced against its time per FRAME.

VERDICT r5: the object divided the bytes of a two-launch stage (`preprocess_fwd` = sh0 kernel + preprocess kernel) by the mean time of ONE
launch and printed 7.1 TB/s = 0.89 of the roof for C5 -- more than the box can fill memory at."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench


def test_one_bracket_per_frame_is_bytes_over_launch_time():
    r = bench.dominant_roofline("render_bwd", 271.3e6, 0.3323, 20, 20)
    assert abs(r["achieved"] - 816.4) < 1.0 and abs(r["frac"] - 0.102) < 1e-3
    assert r["launch_groups_per_frame"] == 1.0 and r["avg_launch_ms"] == 0.3323 and "instrument_error" not in r


def test_two_brackets_per_frame_use_the_stage_time_per_frame():
    # 1.2 GB per frame, two launch groups of 0.10 ms (mean) per frame: 0.20 ms per frame = 6 TB/s, not 12
    r = bench.dominant_roofline("preprocess_fwd", 1.2e9, 0.10, 200, 100)
    assert r["launch_groups_per_frame"] == 2.0 and abs(r["avg_launch_ms"] - 0.20) < 1e-9
    assert abs(r["achieved"] - 6000.0) < 1.0 and "instrument_error" not in r


def test_a_rate_above_the_fill_rate_is_flagged_not_reported():
    r = bench.dominant_roofline("preprocess_fwd", 1.2e9, 0.10, 100, 100)      # what the old arithmetic produced: 12 TB/s
    assert r["achieved"] > bench.HBM_FILL_GBS and r["frac"] is None and "instrument_error" in r


def test_iteration_mode_counts_every_camera_of_a_step():
    # 5 cameras per step, 20 steps, one bracket per camera
    r = bench.dominant_roofline("render_bwd", 500e6, 0.5, 100, 100)
    assert r["launch_groups_per_frame"] == 1.0 and abs(r["achieved"] - 1000.0) < 1e-6
