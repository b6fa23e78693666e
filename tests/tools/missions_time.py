#!/usr/bin/env python3
"""Developer tool: bench.py's closed-loop missions alone (p50 / p99 per replanning step and the stage times)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
import bench  # noqa: E402

if __name__ == "__main__":
    out = bench.closed_loop_latency(0)
    for k, v in out.items():
        if isinstance(v, dict):
            print("%-24s p50 %.4f  p99 %.4f  graph %s stages %s" % (k, v.get("p50_ms", float("nan")), v.get("p99_ms", float("nan")), v.get("p50_ms_as_one_hip_graph"),
                                                         {a: round(b, 4) for a, b in (v.get("stage_p50_ms") or {}).items()}))
