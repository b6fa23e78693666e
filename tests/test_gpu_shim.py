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


@pytest.mark.gpu
def test_cpp_interface_with_octomaps_own_point3d(gpu_required):
    """matp.hpp's MATP_HAVE_OCTOMAP branch -- what a maintainer of the reference compiles -- with
    octomap::point3d from a test double of octomap's header (tests/cpp/octomap_double: octomap is not in this image): the same
    known answers through the same test."""
    d = os.path.join(ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-s", "-C", d, "shim_test_octomap"])
    out = subprocess.run([os.path.join(d, "shim_test_octomap")], capture_output=True, text=True, timeout=120)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shim_test: ok" in out.stdout


def test_the_octomap_branch_of_the_mirror_compiles():
    """CPU tier: the MATP_HAVE_OCTOMAP branch of matp.hpp compiles (and links against the library) -- it had never been compiled."""
    d = os.path.join(ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-s", "-C", d, "shim_test_octomap"])
    assert os.path.exists(os.path.join(d, "shim_test_octomap"))
