"""-m gpu: dlscqp_selftest -- every kernel instantiation a context can hand a batch to plans a wall problem of the
context's shape at creation; a library whose kernels plan wrongly refuses to create a context."""
import ctypes as C
import os
import subprocess
import sys

import pytest

from dlsc_gc_planner_amd import abi, workloads
from dlsc_gc_planner_amd.planner import BatchTrajOptimizer

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cfg", [
    workloads.s64_config(), workloads.x2048_config(),
    abi.make_config(dim=3, M=10, communication_range=3.0, world_use_octomap=True, world_min=(-5.0, -5.0, 0.0), world_max=(5.0, 5.0, 2.5)),
    abi.make_config(dim=3, M=4), abi.make_config(dim=2, M=12, communication_range=3.0), abi.make_config(dim=3, M=5, phi_n=3),
], ids=["s64", "x2048", "forest", "M4", "M12", "phin3"])
def test_selftest_passes_for_the_shapes_of_the_reference(gpu_required, cfg):
    opt = BatchTrajOptimizer(cfg)  # dlscqp_create runs it
    try:
        assert opt.lib.dlscqp_selftest(opt.handle) == abi.SUCCESS
    finally:
        opt.close()


def test_a_kernel_that_plans_wrongly_fails_the_selftest(gpu_required, tmp_path):
    """A build whose plan kernel stops after its first pricing pass (the -DDQ_PHASE_PROFILE profiling build with
    DLSCQP_DEBUG_EXIT_PHASE=2: every agent returns its unconstrained optimum as DLSCQP_OK) must not get a context."""
    lib = os.path.join(ROOT, "dlsc_gc_planner_amd", "csrc", "libdlscqp_phase.so")
    if not os.path.exists(lib):
        pytest.skip("profiling build not present (tests/tools/phase_cost.py builds it)")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from dlsc_gc_planner_amd import workloads\n"
            "from dlsc_gc_planner_amd.planner import BatchTrajOptimizer\n"
            "try:\n    BatchTrajOptimizer(workloads.s64_config())\n    print('CREATED')\n"
            "except Exception as e:\n    print('REFUSED', e)\n" % ROOT)
    env = dict(os.environ, DLSCQP_LIB=lib, DLSCQP_DEBUG_EXIT_PHASE="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "REFUSED" in r.stdout and "selftest failed" in r.stdout, (r.stdout, r.stderr[-500:])
    env = dict(os.environ, DLSCQP_LIB=lib)  # the same build left alone plans correctly
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "CREATED" in r.stdout, (r.stdout, r.stderr[-500:])
