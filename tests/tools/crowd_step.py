#!/usr/bin/env python3
"""Diagnostic: bench.py's x2048 moving crowd alone (for rocprofv3 --kernel-trace: every kernel of the flown step)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
import bench  # noqa: E402

if __name__ == "__main__":
    print(json.dumps(bench.moving_crowd(0)))
