#!/usr/bin/env python3
"""Developer tool: the moving crowd (bench.py: x2048_moving_crowd) with and without helper wavefronts.
    DLSCQP_WPA=1|4 [DLSCQP_LIB=...] python tests/tools/crowd_time.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
import bench  # noqa: E402

if __name__ == "__main__":
    out = bench.moving_crowd(0, heavy_ranks=int(os.environ.get("HEAVY_RANKS", "0")))
    print("heavy_ranks=%s WPA=%s lib=%s: crowd %.4f ms/step, stages %s, iterations mean %.2f max %d" % (
        os.environ.get("HEAVY_RANKS", "0"), os.environ.get("DLSCQP_WPA", "auto"), os.path.basename(os.environ.get("DLSCQP_LIB", "libdlscqp.so")), out["ms_per_step"],
        {k: round(v, 4) for k, v in out["stage_p50_ms"].items()}, out["mean_active_set_iterations"], out["max_active_set_iterations"]))
