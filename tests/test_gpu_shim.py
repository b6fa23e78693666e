"""-m gpu: the reference's C++ interface (MATP::TrajOptimizer / GoalOptimizer / CollisionConstraints,
mirrored in dlsc_gc_planner_amd/include/dlsc_gc_planner_amd/matp.hpp) over the HIP library."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_interface_known_answers_and_error_convention(gpu_required):
    d = os.path.join(ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-s", "-C", d])
    out = subprocess.run([os.path.join(d, "shim_test")], capture_output=True, text=True, timeout=120)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shim_test: ok" in out.stdout
